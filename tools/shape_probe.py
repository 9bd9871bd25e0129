"""Lean scan bandwidth by shape / element type (what limits config 3's 5.7 TB/s?): tuning aid."""
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
cases = [("f32", 1024, 256, 4096), ("f32", 256, 256, 1024), ("bf16", 256, 512, 1024), ("f32", 256, 512, 1024), ("f32", 256, 256, 4096),
         ("f32", 64, 256, 4096), ("bf16", 1024, 512, 1024), ("f32", 1024, 256, 1024)]
with torch.cuda.stream(side):
    for dt, B, D, S in cases:
        bench.WORKLOADS["probe"] = ("paged", B, D, S)
        wl = bench.Workload("probe", dev, 0x5EED, headroom=8, dtype=dt)
        scan = lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, dt == "bf16", phases=7, n_sequence=wl.S)
        alg = wl.algorithmic_bytes(wl.lengths_host)["scan_lean"]
        r = {}
        for nt in (1, 0):
            lib.mli_tune(b"nt_loads", nt)
            us = timed(scan, 50, side); r[f"nt{nt}"] = (round(us, 1), round(alg / us / 1e6, 3))
        out[f"{dt}_B{B}_D{D}_S{S}"] = {"MB": round(alg / 1e6), **r}
        del wl; torch.cuda.empty_cache()
print(json.dumps(out, indent=0))
