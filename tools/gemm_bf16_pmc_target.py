"""The bf16 decode projection at B=1024, D=2048 in its default form (LDS-DMA loaders), 30 launches: the target of the
rocprofv3 --pmc passes behind profiles/r03_gemm_bf16_dma_pmc.json (dev tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = bench.Workload("e1", torch.device("cuda:0"), 123, headroom=8, dtype="bf16")
fn = [v for k, v in wl.kernels().items() if k.startswith("get_latest")][0]
for _ in range(30):
    fn()
torch.cuda.synchronize()
