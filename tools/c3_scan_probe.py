"""Config-3 lean scan: chunk sizes of the chunked grid against equal page shares (tuning aid)."""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from min_llm_inference_amd import load_library, ops
from step_probe import timed
lib = load_library()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
side = torch.cuda.Stream(device=dev)
out = {}
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
with torch.cuda.stream(side):
    wl = bench.Workload(name, dev, 0x5EED, headroom=8, dtype="f32")
    scan = lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, False, phases=7, n_sequence=wl.S)
    alg = wl.algorithmic_bytes(wl.lengths_host)["scan_lean"]
    lib.mli_tune(b"scan_stream", 0)
    for ct in (0, 64, 128, 256, 512):
        lib.mli_tune(b"chunk_tokens", ct)
        us = timed(scan, 200, side); out[f"chunked_ct{ct}"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3)}
    lib.mli_tune(b"chunk_tokens", 0)
    for nt in (1, 0):
        lib.mli_tune(b"nt_loads", nt)
        us = timed(scan, 200, side); out[f"chunked_nt{nt}"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3)}
    lib.mli_tune(b"nt_loads", 2)
    lib.mli_tune(b"scan_stream", 1); lib.mli_tune(b"scan_stream_min_tokens", 0)
    for pct, gran in ((0, 16), (4, 16), (8, 16), (12, 16), (4, 64)):
        lib.mli_tune(b"scan_stream_dynamic_pct", pct); lib.mli_tune(b"scan_stream_granule", gran)
        us = timed(scan, 200, side)
        out[f"stream_dyn{pct}_gran{gran}"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3)}
print(json.dumps(out, indent=1))
