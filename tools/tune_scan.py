"""Sweep the scan kernels' tuning knobs on one workload (dev tool; not part of the product or the bench)."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from min_llm_inference_amd import load_library  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
wl = bench.Workload(name, dev, 123, headroom=8, dtype=dtype)
lib = load_library()
alg = wl.algorithmic_bytes(wl.lengths_host)
ks = wl.kernels()
qk = [k for k in ks if k.startswith("qkt")][0]
sk = [k for k in ks if k.startswith("softmax_v")][0]
print(f"workload {name}: qkt bytes {alg['qkt']/1e9:.3f} GB, sv bytes {alg['softmax_v']/1e9:.3f} GB")
for ct, nt, tb in itertools.product([0, 64, 128, 256, 512, 1024], [1, 0], [8]):
    assert lib.mli_tune(b"chunk_tokens", ct) == 0
    assert lib.mli_tune(b"nt_loads", nt) == 0
    assert lib.mli_tune(b"qkt_token_batch", tb) == 0
    tq = bench.time_kernel(ks[qk], 20)
    line = f"ct {ct:5d} nt {nt} tb {tb:2d}: qkt {tq*1e3:8.1f} us {alg['qkt']/tq/1e6:7.0f} GB/s"
    if tb == 8:
        ts = bench.time_kernel(ks[sk], 20)
        line += f" | sv {ts*1e3:8.1f} us {alg['softmax_v']/ts/1e6:7.0f} GB/s"
    print(line, flush=True)
