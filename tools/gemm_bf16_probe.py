"""bf16 decode projection at B=1024, D=2048: the tiled kernel (split0) against the LDS-DMA loader-wave / MFMA-wave kernel (split1),
timed in one process; checks that pages and q_output are bit-identical.  (profiles/r03_gemm_bf16_probe.json was taken while round
2's register-staging loader kernel still existed: there split1 is that kernel and split2 this one.)"""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from min_llm_inference_amd import load_library, ops
lib = load_library()
dev = torch.device("cuda:0")
wl = bench.Workload("e1", dev, 123, headroom=8, dtype="bf16")
fn = [v for k, v in wl.kernels().items() if k.startswith("get_latest")][0]
flops = 2.0 * wl.B * wl.D * 3 * wl.D
out = {}
for rnd in range(3):
    for split in (0, 1):
        lib.mli_tune(b"gemm_bf16_split", split)
        t = bench.time_kernel(fn, 200)
        out.setdefault(f"split{split}", []).append((round(t * 1e3, 1), round(flops / (t * 1e-3) / 1e12)))
# same results?
lib.mli_tune(b"gemm_bf16_split", 0); fn(); torch.cuda.synchronize(); q0 = wl.q_output.clone(); p0 = wl.pool.clone()
lib.mli_tune(b"gemm_bf16_split", 1); wl.q_output.zero_(); fn(); torch.cuda.synchronize()
out["identical"] = bool(torch.equal(q0, wl.q_output) and torch.equal(p0.view(torch.int16), wl.pool.view(torch.int16)))
print(json.dumps(out))
