"""bf16 decode projection over batch sizes and widths: the LDS-DMA loader kernel (gemm_bf16_split = 1) against the tiled kernels
(0) under the launcher's own heuristics -- where does the 128 x 192-tile kernel, one workgroup per tile, stop paying (few tiles:
idle CUs)?  python tools/gemm_bf16_shape_probe.py"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from min_llm_inference_amd import load_library, ops
from helpers import bf16_bits, build_page_pool
lib = load_library()
dev = torch.device("cuda:0")
out = {}
for B, D in ((700, 2048), (1024, 2048), (2048, 2048), (1024, 1536), (1400, 1024), (2048, 1024), (4096, 1024), (1024, 4096)):
    S = 16
    rng = np.random.default_rng(B + D)
    lengths = rng.integers(1, S, size=B).astype(np.int32)
    pool, table = build_page_pool(rng, lengths, S, D)
    p = torch.from_numpy(bf16_bits(((rng.random(pool.shape, dtype=np.float32) * 2 - 1))).view(np.int16)).to(dev).view(torch.bfloat16)
    t = torch.from_numpy(np.where(table >= 0, p.data_ptr() + 2 * table, 0).astype(np.int64)).to(dev)
    ws = [torch.from_numpy(bf16_bits(((rng.random((D, D), dtype=np.float32) * 2 - 1) / np.sqrt(D)).astype(np.float32)).view(np.int16)).to(dev).view(torch.bfloat16) for _ in range(3)]
    L = torch.from_numpy(lengths).to(dev)
    q = torch.zeros(B, D, device=dev)
    fn = lambda: ops.launch_get_latest_k_q_v_paged_attention_bf16(t, L, ws[0], ws[1], ws[2], q, S)
    r = {}
    for rnd in range(2):
        for split in (0, 1):
            lib.mli_tune(b"gemm_bf16_split", split)
            us = bench.time_kernel(fn, 100) * 1e3
            r.setdefault(f"split{split}_us", []).append(round(us, 1))
    r["tflops_split1"] = round(2.0 * B * D * 3 * D / (min(r["split1_us"]) * 1e-6) / 1e12)
    out[f"B{B}_D{D}"] = r
    del p, t, ws, q
    torch.cuda.empty_cache()
lib.mli_tune(b"gemm_bf16_split", 1)
print(json.dumps(out, indent=1))
