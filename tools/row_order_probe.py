"""One-workgroup-per-row scans (short sequences): rows in grid order against longest first (tuning aid)."""
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
    for name, dt in (("e1", "f32"), ("e1", "bf16")):
        wl = bench.Workload(name, dev, 0x5EED, headroom=8, dtype=dt)
        alg = wl.algorithmic_bytes(wl.lengths_host)["scan_lean"]
        ref = None
        for order in (0, 1, 0, 1):
            lib.mli_tune(b"scan_row_order", order)
            for lean in (True, False):
                scan = (lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, dt == "bf16", phases=7, n_sequence=wl.S)) if lean else \
                       (lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, wl.qkt_output, wl.attention_result, dt == "bf16", phases=3))
                us = timed(scan, 100, side)
                out.setdefault(f"{name}_{dt}_{'lean' if lean else 'materialising'}_order{order}", []).append((round(us, 1), round(alg / us / 1e6, 2)))
                side.synchronize()
                if lean:
                    if ref is None: ref = wl.attention_result.clone()
                    else: assert torch.equal(ref, wl.attention_result)
        del wl; torch.cuda.empty_cache()
print(json.dumps(out))
