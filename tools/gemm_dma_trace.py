"""Cycle budget of the LDS-DMA bf16 decode projection (gemm_bf16_dma_kernel built with -DMLI_DMA_TRACE):

    cd min_llm_inference_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Ihost/include -Icsrc \
        -DMLI_DMA_TRACE -c csrc/proj_gemm_bf16.hip -o /tmp/t.o && hipcc --offload-arch=gfx950 -shared -fPIC \
        -o ../tools/build/libmli_dma_trace.so /tmp/t.o $(ls build/*.o | grep -v proj_gemm_bf16.o) -L/opt/rocm/lib -lrocprofiler-sdk-roctx
    python tools/gemm_dma_trace.py tools/build/libmli_dma_trace.so
A tuning aid, never the product."""
import ctypes, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from min_llm_inference_amd import _lib
path = os.path.abspath(sys.argv[1]); _lib.library_path = lambda: path
import bench
wl = bench.Workload("e1", torch.device("cuda:0"), 123, headroom=8, dtype="bf16")
fn = [v for k, v in wl.kernels().items() if k.startswith("get_latest")][0]
for _ in range(5):
    fn()
torch.cuda.synchronize()
t_us = bench.time_kernel(fn, 200) * 1e3
fn(); torch.cuda.synchronize()
raw = ctypes.CDLL(path)
buf = np.zeros((1024, 16), np.uint64)
assert raw.mli_debug_dma_trace(buf.ctypes.data_as(ctypes.c_void_p), 1024) == 0
t = buf[buf[:, 4] > 0].astype(np.float64)
# clock64() counts per XCD (the counters of different XCDs are not aligned): only differences inside a workgroup mean anything
out = {"workgroups": len(t), "launch_us": round(t_us, 1)}
def us(x): return round(float(x), 0)
out["per_workgroup_ticks"] = {
    "entry -> rows resolved": us(np.mean(t[:, 1] - t[:, 0])),
    "rows resolved -> first tile readable": us(np.mean(t[:, 2] - t[:, 1])),
    "k loop": us(np.mean(t[:, 3] - t[:, 2])),
    "k loop end -> stores landed": us(np.mean(t[:, 4] - t[:, 3])),
    "  of which: barrier E1 / accumulators to LDS + E2 / reads + store issue / stores landing":
        [us(np.mean(t[:, 11] - t[:, 3])), us(np.mean(t[:, 12] - t[:, 11])), us(np.mean(t[:, 13] - t[:, 12])), us(np.mean(t[:, 4] - t[:, 13]))],
    "lifetime": us(np.mean(t[:, 4] - t[:, 0])),
    "lifetime p0 / p50 / p100": [us(np.min(t[:, 4] - t[:, 0])), us(np.median(t[:, 4] - t[:, 0])), us(np.max(t[:, 4] - t[:, 0]))],
}
out["clock64_ticks_per_ns"] = round(float(np.mean((t[:, 4] - t[:, 0]) / (t[:, 6] * 10.0))), 3)   # wall_clock64: 100 MHz
loop = np.mean(t[:, 3] - t[:, 2])
out["k_loop_shares"] = {
    "MFMA wave 0: inside the barrier": round(float(np.mean(t[:, 5]) / loop), 3),
    "loader wave 4: waiting for its tile (vmcnt)": round(float(np.mean(t[:, 8]) / loop), 3),
    "loader wave 4: inside the barrier": round(float(np.mean(t[:, 9]) / loop), 3),
    "loader wave 4: issuing the next tile": round(float(np.mean(t[:, 10]) / loop), 3),
}
out["k_loop_ticks_per_step"] = round(float(loop) / (wl.D // 64), 1)
print(json.dumps(out, indent=1))
