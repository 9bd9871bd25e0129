"""Single-pass scan vs the three separate kernels on one workload (dev tool): python tools/tune_wide.py e1 f32"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "e1"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
wl = bench.Workload(name, dev, 123, headroom=8, dtype=dtype)
alg = wl.algorithmic_bytes(wl.lengths_host)
print(f"workload {name} {dtype}: scan bytes {alg['scan']/1e9:.3f} GB")
for rnd in range(2):
    for label, ks in (("fused", wl.kernels()), ("separate", wl.kernels_separate())):
        tot = 0.0
        for k, fn in ks.items():
            t = bench.time_kernel(fn, 20)
            print(f"  {label:9s} {k[:60]:60s} {t*1e3:8.1f} us", flush=True)
            if not k.startswith("get_latest"):
                tot += t
        print(f"{label}: scan total {tot*1e3:.1f} us = {alg['scan']/tot/1e6:.0f} GB/s", flush=True)

from min_llm_inference_amd import load_library  # noqa: E402
lib = load_library()
for ct in (128, 64, 0):
    lib.mli_tune(b"chunk_tokens", ct)
    ks = wl.kernels()
    tot = sum(bench.time_kernel(fn, 20) for k, fn in ks.items() if not k.startswith("get_latest"))
    print(f"chunk_tokens {ct}: fused scan + combine {tot*1e3:.1f} us = {alg['scan']/tot/1e6:.0f} GB/s", flush=True)
