"""One-launch decode step against the separate launches (config 3 and smaller batches): tuning aid."""
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
shapes = {"c3": None, "b128": ("paged", 128, 256, 1024), "b512_s512": ("paged", 512, 256, 512), "b64_d512": ("paged", 64, 512, 2048)}
with torch.cuda.stream(side):
    for name, shape in shapes.items():
        if shape is not None:
            bench.WORKLOADS[name] = shape
        wl = bench.Workload(name, dev, 0x5EED, headroom=8, dtype="f32")
        L0 = wl.lengths.clone()
        for fused in (1, 0, 1, 0):
            lib.mli_tune(b"step_fused", fused)
            wl.lengths.copy_(L0)
            us = timed(wl.lean_step, 100, side)
            out.setdefault(name, {}).setdefault(f"fused{fused}", []).append(round(us, 1))
        del wl; torch.cuda.empty_cache()
lib.mli_tune(b"step_fused", 0)
print(json.dumps(out))
