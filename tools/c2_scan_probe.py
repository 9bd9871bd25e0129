"""Config-2 lean step and the contiguous single-launch scan on their own (tuning aid)."""
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
    wl = bench.Workload("c2", dev, 0x5EED, headroom=8, dtype="f32")
    L0 = wl.lengths.clone()
    alg = wl.algorithmic_bytes(wl.lengths_host)["scan_lean"]
    scan = lambda: ops.decode_scan_contiguous(wl.q_output, wl.kt_cache, wl.v_cache, wl.lengths, wl.attention_result)
    for nt in (0, 1):
        lib.mli_tune(b"nt_loads", nt)
        us = timed(scan, 200, side); out[f"scan_nt{nt}"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3)}
    lib.mli_tune(b"nt_loads", 2)
    us = timed(wl.lean_step, 100, side); out["lean_step_us"] = round(us, 1); wl.lengths.copy_(L0)
    us = timed(wl.step, 100, side); out["materialising_step_us"] = round(us, 1)
print(json.dumps(out))
