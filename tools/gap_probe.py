"""Lean decode steps back to back, nothing else: for a rocprofv3 --kernel-trace run (tools/gap_report.py reads the CSV)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    wl = bench.Workload(name, dev, 0x5EED, headroom=80, dtype="bf16" if name == "c4" else "f32")
    for _ in range(60):
        wl.lean_step()
    side.synchronize()
