"""A/B the composition forms (flash_decode on/off, chunk sizes) in one process (dev tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from min_llm_inference_amd import load_library  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
wl = bench.Workload(name, dev, 123, headroom=8, dtype=dtype)
lib = load_library()
alg = wl.algorithmic_bytes(wl.lengths_host)
print(f"workload {name} {dtype}: step attention bytes {alg['step']/1e9:.3f} GB")
scan = [v for k, v in wl.kernels().items() if k.startswith("fused_decode_scan")][0]
for rnd in range(3):
    for dyn in (1, 0):
        for ct in (0, 1024):
            lib.mli_tune(b"scan_dynamic_items", dyn)
            lib.mli_tune(b"chunk_tokens", ct)
            t = bench.time_kernel(scan, 20)
            print(f"round {rnd} dynamic items {dyn} ct {ct:5d}: scan {t*1e3:8.1f} us  {alg['scan']/t/1e6:7.0f} GB/s", flush=True)
