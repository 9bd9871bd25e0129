"""Timeline of the one-launch decode step (decode_step_fused.hip built with -DMLI_STEP_TRACE):
    python tools/step_fused_trace.py <trace lib.so> [workload]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from min_llm_inference_amd import _lib
path = os.path.abspath(sys.argv[1]); _lib.library_path = lambda: path
import bench
name = sys.argv[2] if len(sys.argv) > 2 else "c3"
lib = _lib.load_library()
lib.mli_tune(b"step_fused", 1)
wl = bench.Workload(name, torch.device("cuda:0"), 123, headroom=24, dtype="f32")
for _ in range(5):
    wl.lean_step()
torch.cuda.synchronize()
raw = ctypes.CDLL(path)
raw.mli_debug_step_trace_clear()
wl.lean_step()
torch.cuda.synchronize()
n = 8192
buf = np.zeros((n, 8), np.uint64)
assert raw.mli_debug_step_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
t = buf[buf[:, 0] > 0].astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0
role = t[:, 6]
print(f"{len(t)} workgroups; span {us(np.maximum(t[:, 4], t[:, 7]).max()):.1f} us")
for r, nm in ((1, "projection"), (2, "scan"), (3, "logits")):
    m = role == r
    if not m.any(): continue
    x = t[m]
    pct = lambda v: np.round(np.percentile(v, [0, 50, 95, 100]), 1)
    print(f" {nm}: {m.sum()} wgs; entry {pct(us(x[:, 0]))}; ticket latency {pct((x[:, 1] - x[:, 0]) / 100.0)}; end {pct(us(x[:, 4]))}")
    w = x[x[:, 3] > 0]
    if len(w):
        print(f"    reached the gate at {pct(us(w[:, 2]))}, waited {pct((w[:, 3] - w[:, 2]) / 100.0)}, gate -> end {pct((w[:, 4] - w[:, 3]) / 100.0)} ({len(w)} wgs)")
    if r == 2:
        e = x[x[:, 3] == 0]
        print(f"    empty items: {len(e)}, entry -> end {pct((e[:, 4] - e[:, 0]) / 100.0)}")
    if r == 3:
        f = x[x[:, 7] > 0]
        print(f"    finalizers: {len(f)}; tile done -> finalized {pct((f[:, 7] - f[:, 4]) / 100.0)}; finalized at {pct(us(f[:, 7]))}")
edges = np.linspace(0, us(np.maximum(t[:, 4], t[:, 7]).max()), 13)
print(" window (us): workgroups entering / scan items past the gate in flight")
for lo, hi in zip(edges[:-1], edges[1:]):
    ent = ((us(t[:, 0]) >= lo) & (us(t[:, 0]) < hi)).sum()
    s = t[(role == 2) & (t[:, 3] > 0)]
    mid = (lo + hi) / 2
    infl = ((us(s[:, 3]) <= mid) & (us(s[:, 4]) > mid)).sum()
    print(f"  {lo:6.1f}-{hi:6.1f}: {ent:5d}  {infl:5d}")
