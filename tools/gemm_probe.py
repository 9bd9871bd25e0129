"""Run only the decode projection GEMM of a workload a few times (dev tool, for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "e1"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
wl = bench.Workload(name, torch.device("cuda:0"), 123, headroom=8, dtype=dtype)
fn = [v for k, v in wl.kernels().items() if k.startswith("get_latest")][0]
t = bench.time_kernel(fn, 20)
flops = 2.0 * wl.B * wl.D * 3 * wl.D
print(f"{name} {dtype}: projection {t*1e3:.1f} us = {flops/(t*1e-3)/1e12:.1f} TFLOP/s")
from min_llm_inference_amd import load_library  # noqa: E402
lib = load_library()
for rnd in range(3):
    for tall in (0, 1):
        lib.mli_tune(b"gemm_tall_tiles", tall)
        lib.mli_tune(b"gemm_deep_k", tall)
        t = bench.time_kernel(fn, 200)
        print(f"round {rnd} gemm_tall_tiles=gemm_deep_k={tall}: projection {t*1e3:.1f} us = {flops/(t*1e-3)/1e12:.1f} TFLOP/s", flush=True)
lib.mli_tune(b"gemm_tall_tiles", 1)
lib.mli_tune(b"gemm_deep_k", 1)
for compact in (0, 1, 0, 1):
    lib.mli_tune(b"latest_compact", compact)
    t = bench.time_kernel(fn, 200)
    print(f"latest_compact={compact}: projection {t*1e3:.1f} us", flush=True)
for deep in (0, 1, 0, 1):
    lib.mli_tune(b"gemm_deep_k", deep)
    def decoder_once():   # the decoder advances the lengths: rewind, or 200 calls run past the allocated pages
        wl.lengths.copy_(wl.lengths0)
        wl.decoder()
    t = bench.time_kernel(decoder_once, 200)
    wl.lengths.copy_(wl.lengths0)
    t2 = bench.time_kernel(fn, 200)
    print(f"gemm_deep_k={deep}: decoder head (logits GEMM + argmax) {t*1e3:.1f} us, projection {t2*1e3:.1f} us", flush=True)
