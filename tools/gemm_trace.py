"""Cycle budget of the bf16 projection GEMM's k loop (proj_gemm_bf16.hip built with -DMLI_GEMM_TRACE):
    python tools/gemm_trace.py <trace lib.so> [workload]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from min_llm_inference_amd import _lib
path = os.path.abspath(sys.argv[1]); _lib.library_path = lambda: path
import bench
name = sys.argv[2] if len(sys.argv) > 2 else "e1"
wl = bench.Workload(name, torch.device("cuda:0"), 123, headroom=8, dtype="bf16")
fn = [v for k, v in wl.kernels().items() if k.startswith("get_latest")][0]
for _ in range(5):
    fn()
torch.cuda.synchronize()
raw = ctypes.CDLL(path)
buf = np.zeros((1024, 8), np.uint64)
assert raw.mli_debug_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
t = buf[buf[:, 7] > 0].astype(np.float64)
nk = t[0, 7]
print(f"{len(t)} workgroups traced, {int(nk)} k steps each; clock64 ticks (100 MHz: 1 tick = 10 ns)")
names = ["issue next tile's loads", "fragment reads + MFMAs", "barrier 1", "wait for the loads + store to LDS", "barrier 2"]
tot = t[:, 5]
for i, n in enumerate(names):
    print(f"  {n:36s} mean {t[:, i].mean():9.0f} ticks = {100 * t[:, i].mean() / tot.mean():5.1f} % of the loop   per step {t[:, i].mean() / nk:7.1f}")
print(f"  whole k loop: mean {tot.mean():.0f} ticks = {tot.mean() / 100:.1f} us; start spread {(t[:, 6].max() - t[:, 6].min()) / 100:.1f} us")
