"""Race screen of the LDS-DMA bf16 decode projection (gemm_bf16_dma_kernel: counted vmcnt + raw barriers + LDS-DMA into stages
that are re-used two tiles later): many launches per shape, every one compared bit for bit with the tiled kernel's pages and
q_output.  A wrong wait count or barrier shows up as rare wrong tiles that a single run of the parity test can miss.
    python tools/gemm_bf16_race_screen.py [launches per shape] [load]
With `load`, a second stream streams a 1 GiB buffer through the chip meanwhile: DMA landings arrive late and out of step."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from min_llm_inference_amd import load_library, ops
from helpers import bf16_bits, build_page_pool
lib = load_library()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
load = len(sys.argv) > 2 and sys.argv[2] == "load"
side = torch.cuda.Stream(device=dev)
big = torch.empty(1 << 28, device=dev).uniform_() if load else None
sink = torch.zeros(64, device=dev)
out = {"under_load": load}
lib.mli_tune(b"gemm_tall_tiles", 2)
for B, S, D in ((1024, 32, 2048), (1000, 16, 1024), (300, 16, 1088), (2048, 16, 1152)):
    rng = np.random.default_rng(B + D)
    lengths = rng.integers(0, S, size=B).astype(np.int32)
    lengths[::7] = 0
    pool, table = build_page_pool(rng, lengths, S, D)
    pool = ((rng.random(pool.shape, dtype=np.float32) * 2 - 1)).astype(np.float32)
    w = [((rng.random((D, D), dtype=np.float32) * 2 - 1) / np.sqrt(D)).astype(np.float32) for _ in range(3)]
    bits0 = torch.from_numpy(bf16_bits(pool).view(np.int16)).to(dev)
    ws = [torch.from_numpy(bf16_bits(x).view(np.int16)).to(dev).view(torch.bfloat16) for x in w]
    L = torch.from_numpy(lengths).to(dev)

    def run(split):
        lib.mli_tune(b"gemm_bf16_split", split)
        p = bits0.clone().view(torch.bfloat16)
        t = torch.from_numpy(np.where(table >= 0, p.data_ptr() + 2 * table, 0).astype(np.int64)).to(dev)
        q = torch.full((B, D), 7.0, device=dev)
        ops.launch_get_latest_k_q_v_paged_attention_bf16(t, L, ws[0], ws[1], ws[2], q, S)
        return p.view(torch.int16), q

    p_ref, q_ref = run(0)
    bad = 0
    for i in range(n):
        if load and i % 4 == 0:
            with torch.cuda.stream(side):
                ops.stream_read(big, sink)
        p, q = run(1)
        if not (torch.equal(p, p_ref) and torch.equal(q, q_ref)):
            bad += 1
    out[f"B{B}_D{D}"] = {"launches": n, "mismatching_launches": bad}
lib.mli_tune(b"gemm_tall_tiles", 1); lib.mli_tune(b"gemm_bf16_split", 1)
print(json.dumps(out))
sys.exit(1 if any(v["mismatching_launches"] for v in out.values() if isinstance(v, dict)) else 0)
