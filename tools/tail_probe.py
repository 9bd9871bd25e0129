import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from min_llm_inference_amd import load_library, ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from step_probe import timed
lib = load_library()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
side = torch.cuda.Stream(device=dev)
out = {}
with torch.cuda.stream(side):
    wl = bench.Workload("c4", dev, 0x5EED, headroom=8, dtype="bf16")
    scan = lambda: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths, None, wl.attention_result, True, phases=7, n_sequence=wl.S)
    for ct in (512, 1024):
        lib.mli_tune(b"chunk_tokens", ct if ct != 512 else 0)
        for tail in (64, 128, 256, 512, 1024):
            if tail > ct: continue
            lib.mli_tune(b"scan_tail_tokens", tail)
            ts = [timed(scan, 50, side) for _ in range(3)]
            out[f"ct{ct}_tail{tail}"] = [round(t, 1) for t in ts]
    lib.mli_tune(b"scan_tail_tokens", 0); lib.mli_tune(b"chunk_tokens", 0)
    lib.mli_tune(b"scan_partial_last", 0)
    out["plain_order"] = [round(timed(scan, 50, side), 1) for _ in range(3)]
    lib.mli_tune(b"scan_partial_last", 1)
    alg = wl.algorithmic_bytes(wl.lengths_host)
    out["alg_scan_lean_bytes"] = alg["scan_lean"]
print(json.dumps(out, indent=1))
