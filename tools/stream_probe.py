"""Equal-page-shares scan (attention_stream.hip) against the chunked grid, lean form, same state: a tuning aid."""
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
with torch.cuda.stream(side):
    for name, dt in (("c4", "bf16"),):
        wl = bench.Workload(name, dev, 0x5EED, headroom=8, dtype=dt)
        scan = lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, dt == "bf16", phases=7, n_sequence=wl.S)
        alg = wl.algorithmic_bytes(wl.lengths_host)["scan_lean"]
        lib.mli_tune(b"scan_stream", 0)
        us = timed(scan, 40, side); out[f"{name}_{dt}_chunked"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3)}
        lib.mli_tune(b"scan_stream", 1)
        for pct in (0, 4, 8, 12, 16, 24, 32, 48):
            for gran in (32, 64, 128):
                if pct == 0 and gran != 32: continue
                lib.mli_tune(b"scan_stream_dynamic_pct", pct); lib.mli_tune(b"scan_stream_granule", gran)
                us = timed(scan, 40, side)
                out[f"{name}_{dt}_dyn{pct}_gran{gran}"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3)}
        lib.mli_tune(b"scan_stream_dynamic_pct", 4); lib.mli_tune(b"scan_stream_granule", 64)
        del wl; torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
