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
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): fn()
torch.cuda.synchronize(); us_per_launch = (time.perf_counter() - t0) / 50 * 1e6
fn(); torch.cuda.synchronize()
buf = np.zeros((1024, 8), np.uint64)
assert raw.mli_debug_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
# per workgroup: [0..4] clocks in the five phases of the k loop (thread 0), [5] the k loop, [6] its start stamp, [7] k steps
t = buf[buf[:, 5] > 0].astype(np.float64)
print(f"{len(t)} workgroups traced; ~{us_per_launch:.1f} us per launch (host clock, back to back)")
names = ["issue next tile's loads", "fragment reads + MFMAs", "barrier 1", "wait for the loads + store to LDS", "barrier 2"]
tot = t[:, 5]
for i, n in enumerate(names):
    print(f"  {n:36s} {100 * t[:, i].mean() / tot.mean():5.1f} % of the k loop")
print(f"  k loop: {tot.mean():.0f} clocks = {np.mean(tot / np.maximum(t[:, 7], 1)):.0f} per k step of the staged tile")
