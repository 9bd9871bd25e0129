"""Per-workgroup timeline of the equal-page-shares scan (attention_stream.hip built with -DMLI_SCAN_TRACE).
    python tools/stream_trace.py tools/libmli_trace.so [workload] [dtype]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from min_llm_inference_amd import _lib
path = os.path.abspath(sys.argv[1]); _lib.library_path = lambda: path
import bench
from min_llm_inference_amd import ops
name = sys.argv[2] if len(sys.argv) > 2 else "c4"
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
lib = _lib.load_library()
lib.mli_tune(b"scan_stream_min_tokens", 0)
if len(sys.argv) > 4:
    lib.mli_tune(b"scan_stream_dynamic_pct", int(sys.argv[4]))
wl = bench.Workload(name, torch.device("cuda:0"), 123, headroom=8, dtype=dtype)
scan = lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, dtype == "bf16", phases=7, n_sequence=wl.S)
for _ in range(5):
    scan()
torch.cuda.synchronize()
raw = ctypes.CDLL(path)
buf = np.zeros((1024, 8), np.uint64)
assert raw.mli_debug_stream_trace(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
t = buf[buf[:, 4] > 0].astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: x / 100.0
print(f"{len(t)} workgroups, span {us(t[:, 4].max() - t0):.1f} us; pages per share {t[:, 5].mean():.1f}, rows per share {t[:, 7].mean():.2f} (max {t[:, 7].max()})")
for label, a, b in (("entry -> share known, pointers requested", 0, 1), ("-> first page consumed", 1, 2), ("first page -> stream end (wave 0)", 2, 3),
                    ("stream end -> exit (merge, publish)", 3, 4), ("whole workgroup", 0, 4)):
    d = us(t[:, b] - t[:, a])
    print(f"  {label:42s} mean {d.mean():7.2f}  p5 {np.percentile(d, 5):7.2f}  p50 {np.median(d):7.2f}  p95 {np.percentile(d, 95):7.2f}  max {d.max():7.2f}")
print("  starts (us):", np.round(np.percentile(us(t[:, 0] - t0), [0, 50, 100]), 1), " stream ends:", np.round(np.percentile(us(t[:, 3] - t0), [0, 5, 50, 95, 100]), 1),
      " exits:", np.round(np.percentile(us(t[:, 4] - t0), [0, 5, 50, 95, 100]), 1))
xcc = t[:, 6] & 0xF
print("  per XCD: stream end median / max:", [(int(x), round(float(np.median(us(t[xcc == x, 3] - t0)))), round(float(us(t[xcc == x, 3] - t0).max()))) for x in np.unique(xcc)])
rate = t[:, 5] * 16 * 2 * wl.D * wl.esize / np.maximum(us(t[:, 3] - t[:, 1]), 1e-9) / 1e3   # GB/s per workgroup
print(f"  per-workgroup streaming rate GB/s: mean {rate.mean():.1f}  p5 {np.percentile(rate, 5):.1f}  p95 {np.percentile(rate, 95):.1f}")
