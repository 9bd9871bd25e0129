"""fp32 GEMM shapes with few 64x64 tiles per CU: the tiled kernel against its loader / MFMA wave split (tuning aid)."""
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
  for split in (0, 1, 0, 1):
    lib.mli_tune(b"gemm_split", split)
    for name in ("e1", "c4"):
        wl = bench.Workload(name, dev, 0x5EED, headroom=8, dtype="f32")
        r = out.setdefault(f"split{split}", {})
        r.setdefault(f"{name}_logits_fused_head_us", []).append(round(timed(wl.fused_decoder, 50, side), 1)); wl.lengths.copy_(wl.lengths0)
        if name == "e1":   # prefill of n new rows with 32-token prompts
            for n_new in (4, 8, 16, 32, 64, 128):
                idx = torch.arange(n_new, dtype=torch.int32, device=dev)
                L = wl.lengths.clone(); L[:n_new] = 32
                r.setdefault(f"e1_fill_{n_new}x32_us", []).append(round(timed(lambda: ops.launch_fill_new_k_v_cache_paged_attention(wl.page_table, idx, L, wl.wk, wl.wv, n_new, wl.S), 50, side), 1))
        del wl; torch.cuda.empty_cache()
print(json.dumps(out))
