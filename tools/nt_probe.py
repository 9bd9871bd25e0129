"""Non-temporal vs default-policy K/V loads by working-set size (lean scan, fp32, D=256, S=1024; c2 step): tuning aid."""
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
    for B in (64, 128, 256, 512, 1024, 2048):
        bench.WORKLOADS["probe"] = ("paged", B, 256, 1024)
        wl = bench.Workload("probe", dev, 0x5EED, headroom=8, dtype="f32")
        scan = lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, False, phases=7, n_sequence=wl.S)
        alg = wl.algorithmic_bytes(wl.lengths_host)["scan_lean"]
        for nt in (1, 0):
            lib.mli_tune(b"nt_loads", nt)
            us = timed(scan, 100, side); out[f"B{B}_nt{nt}"] = {"us": round(us, 1), "TBps": round(alg / us / 1e6, 3), "MB": round(alg / 1e6)}
        del wl; torch.cuda.empty_cache()
    for name in ("c2", "c3"):
        wl = bench.Workload(name, dev, 0x5EED, headroom=8, dtype="f32")
        L0 = wl.lengths.clone()
        for nt in (1, 0):
            lib.mli_tune(b"nt_loads", nt)
            wl.lengths.copy_(L0)
            us = timed(wl.lean_step, 100, side); out[f"{name}_step_nt{nt}"] = round(us, 1)
        del wl; torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
