"""Where does the single-pass scan spend its time?  Needs a library built with -DMLI_SCAN_TRACE (see
attention_fused.hip); reads the per-workgroup timestamps (100 MHz wall clock) and prints phase statistics and the
busy share of the CU slots.  Dev tool:  python tools/scan_trace.py [lib.so] [workload] [dtype]

    cd min_llm_inference_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Ihost/include -Icsrc \
        -DMLI_SCAN_TRACE -c csrc/attention_fused.hip -o build/attention_fused.o && make     # then copy lib/libmli_hip.so
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from min_llm_inference_amd import _lib  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    path = os.path.abspath(sys.argv.pop(1))
    _lib.library_path = lambda: path
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
lib = _lib.load_library()
wl = bench.Workload(name, torch.device("cuda:0"), 123, headroom=8, dtype=dtype)
scan = [v for k, v in wl.kernels().items() if k.startswith("fused_decode_scan")][0]
for _ in range(5):
    scan()
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.library_path())
raw.mli_debug_scan_trace_clear()
scan()
torch.cuda.synchronize()
n = 16384
buf = np.zeros((n, 8), np.uint64)
assert raw.mli_debug_scan_trace(buf.ctypes.data_as(ctypes.c_void_p), n) == 0
live = buf[:, 4] > 0                       # workgroups that ran to the end (non-empty chunks)
t = buf[live].astype(np.int64)
t0 = t[:, 0].min()


def us(x):                                 # 100 MHz -> microseconds
    return x / 100.0


print(f"{live.sum()} non-empty workgroups, kernel span {us(t[:, 4].max() - t0):.1f} us")
for label, a, b in (("entry -> page pointers in LDS", 0, 1), ("pointers -> first page consumed", 1, 2),
                    ("first page -> last page consumed", 2, 3), ("merge + partial store", 3, 4), ("whole workgroup", 0, 4)):
    d = us(t[:, b] - t[:, a])
    print(f"  {label:36s} mean {d.mean():7.2f} us  p50 {np.median(d):7.2f}  p95 {np.percentile(d, 95):7.2f}")
pages = t[:, 7]
loop = us(t[:, 3] - t[:, 1])
print(f"  pages per workgroup: mean {pages.mean():.1f}; loop time per page of the workgroup: "
      f"{(loop / np.maximum(pages, 1)).mean():.3f} us")
# CU slots: (xcc, se, cu) from HW_ID / XCC_ID
hw, xcc = t[:, 5], t[:, 6] & 0xF
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + cu
span = us(t[:, 4].max() - t0)
busy = []
for k in np.unique(key):
    rows = t[key == k]
    busy.append(us((rows[:, 4] - rows[:, 0]).sum()) / (2 * span))   # two workgroups fit a CU
busy = np.array(busy)
print(f"  {len(busy)} CUs seen; workgroup-time / (2 slots x span): mean {busy.mean():.3f}  min {busy.min():.3f}  max {busy.max():.3f}")
print("  first starts", np.round(us(np.sort(t[:, 0])[:5] - t0), 1), " last ends", np.round(us(np.sort(t[:, 4])[-5:] - t0), 1))
ends = us(np.sort(t[:, 4]) - t0)
print(f"  time at which 90 / 95 / 99 / 100 % of the workgroups have finished: "
      f"{ends[int(0.90 * len(ends))]:.0f} / {ends[int(0.95 * len(ends))]:.0f} / {ends[int(0.99 * len(ends))]:.0f} / {ends[-1]:.0f} us")
# bytes moved per unit of workgroup time, by chunk fill
tok = pages * 16
full = pages >= pages.max()
d_all = us(t[:, 4] - t[:, 0])
print(f"  full chunks: {full.sum()} workgroups, mean {d_all[full].mean():.1f} us each; partial chunks: {(~full).sum()}, "
      f"mean {d_all[~full].mean():.1f} us, mean {tok[~full].mean():.0f} tokens")

# residency over time: how many (non-empty) workgroups a CU holds, chip-wide average per window
edges = np.linspace(0, span, 15)
print("  window (us)      mean workgroups resident per CU   share of CU-time with 2 / 1 / 0 resident")
starts, ends_ = us(t[:, 0] - t0), us(t[:, 4] - t0)
grid_t = np.linspace(0, span, 1400)
res = np.zeros((len(np.unique(key)), len(grid_t)), np.int16)
for i, k in enumerate(np.unique(key)):
    m = key == k
    for s_, e_ in zip(starts[m], ends_[m]):
        res[i, (grid_t >= s_) & (grid_t < e_)] += 1
for lo, hi in zip(edges[:-1], edges[1:]):
    w = (grid_t >= lo) & (grid_t < hi)
    r = res[:, w]
    print(f"  {lo:6.0f} - {hi:6.0f}   {r.mean():.2f}      {(r >= 2).mean():.2f} / {(r == 1).mean():.2f} / {(r == 0).mean():.2f}")
per_xcc = []
for x in np.unique(xcc):
    m = xcc == x
    per_xcc.append((int(x), int(m.sum()), us((t[m, 4] - t[m, 0]).sum()), us(t[m, 4].max() - t0)))
print("  per XCD (id, workgroups, workgroup-time us, last end us):", [(a, b, round(c), round(d)) for a, b, c, d in per_xcc])

# bandwidth over time: a workgroup's bytes (K and V rows of its pages) spread evenly over its streaming interval
row_bytes = wl.D * wl.esize
wg_bytes = pages.astype(np.float64) * 16 * 2 * row_bytes          # whole pages (the last one may be partly dead)
t_begin, t_end = us(t[:, 1] - t0), us(t[:, 3] - t0)
bw = np.zeros(len(grid_t))
dt = grid_t[1] - grid_t[0]
for s_, e_, nb in zip(t_begin, t_end, wg_bytes):
    m = (grid_t >= s_) & (grid_t < e_)
    if m.any():
        bw[m] += nb / (m.sum() * dt)
print("  window (us)      approx. TB/s consumed")
for lo, hi in zip(edges[:-1], edges[1:]):
    w = (grid_t >= lo) & (grid_t < hi)
    print(f"  {lo:6.0f} - {hi:6.0f}   {bw[w].mean() / 1e6:.2f}")
