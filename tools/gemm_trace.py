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
buf = np.zeros((1024, 8), np.uint64); end = np.zeros(1024, np.uint64)
assert raw.mli_debug_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p), end.ctypes.data_as(ctypes.c_void_p), 1024) == 0
m = buf[:, 5] > 0
t = buf[m].astype(np.float64); e = end[m].astype(np.float64)
span = e.max() - t[:, 7].min()
print(f"{len(t)} workgroups traced; kernel span {span:.0f} ticks for ~{us_per_launch:.1f} us per launch (host clock, back to back) -> {span / us_per_launch / 1e3:.2f} ticks per ns")
names = ["issue next tile's loads", "fragment reads + MFMAs", "barrier 1", "wait for the loads + store to LDS", "barrier 2"]
tot = t[:, 5]
for i, n in enumerate(names):
    print(f"  {n:36s} {100 * t[:, i].mean() / tot.mean():5.1f} % of the k loop")
life = e - t[:, 7]
print(f"  per workgroup: entry -> k loop {np.mean(t[:, 6] - t[:, 7]):.0f} ticks, k loop {tot.mean():.0f}, k loop end -> last store landed {np.mean(e - t[:, 6] - tot):.0f}; lifetime {life.mean():.0f} = {100 * life.mean() / span:.0f} % of the span")
print(f"  workgroup entries (ticks after the first): p0 {0:.0f} p50 {np.median(t[:, 7] - t[:, 7].min()):.0f} p100 {(t[:, 7] - t[:, 7].min()).max():.0f}")
