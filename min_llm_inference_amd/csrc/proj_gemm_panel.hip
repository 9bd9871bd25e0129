// Latency-shaped fp32 MFMA GEMM for the SMALL dense products of a decode step -- the projection x[B,D].[Wk|Wq|Wv] and
// the decoder logits attn[B,D].emb[V,D]^T at emb_dim <= 512 and a few hundred rows (BASELINE configs 2 and 3: 0.1 GFLOP,
// 1 MB of operands).  The tiled kernel of proj_gemm.hip runs such a shape as ~50 workgroups of 64x64 that walk K in
// 32-deep steps, every step a global-load round trip, with a quarter of the chip's SIMDs doing all the MFMAs: 11-12 us
// for work the matrix cores finish in under one.  Here
//   * a workgroup owns a 32x32 output tile (4 waves of 16x16, v_mfma_f32_16x16x4_f32): 4x the workgroups, a quarter of
//     the MFMA chain per wave;
//   * the whole K extent of both operand panels (up to 256 k per panel, panels double-buffered through registers) is
//     requested AT ONCE -- one memory round trip instead of K/32 --, staged in LDS as [row][k] with a row pitch of
//     k + 2 floats (fragment reads of 16 rows x 4 k hit 64 different banks; stores are 8-byte aligned b64 pairs);
//   * no row compaction pass and no index build: a row without work simply contributes zeros.
// Numerics: one accumulator chain per output element, k ascending -- bit for bit the k-ordered fmaf chain the 32x32x2
// kernel produces, so the prefill (tiled kernel) and the decode projection (this one) still write identical K / V rows,
// which the engines rely on when a preempted row is re-prefilled.
// Replaces, for those shapes, the same reference functions as proj_gemm.hip (get_latest_kt_q_v, …_paged_attention,
// …_cublas, gemm_transpose_kernel / cublasSgemm of the decoder).
#include <atomic>

#include "gemm_panel_body.hpp"

namespace mli {

template <int MODE, bool BT>
__global__ __launch_bounds__(kPanelThreads) void gemm_f32_panel_kernel(GemmArgs g) {
    extern __shared__ __align__(16) unsigned char panel_smem[];
    gemm_panel_tile<MODE, BT>(g, blockIdx.x, blockIdx.y, panel_smem);
}

static thread_local int g_gemm_panel = 1;  // mli_tune "gemm_panel": 0 = never, 1 = for small grids (default), 2 = whenever the shape allows
void set_gemm_panel(int v) { g_gemm_panel = v < 0 ? 0 : (v > 2 ? 2 : v); }

// The shapes this kernel is for: float4-aligned operands, and so few 64x64 tiles that the tiled kernel would leave most
// of the chip idle while it pays K / 32 dependent load round trips.
bool gemm_panel_wanted(int M, int N_total, int K, bool vec4) {
    if (!vec4 || K % 4 != 0 || g_gemm_panel == 0) return false;
    if (g_gemm_panel == 2) return true;
    const int64_t tiles64 = (int64_t)ceil_div_i(M, 64) * ceil_div_i(N_total, 64);
    return K <= 512 && tiles64 < 256;
}
int gemm_panel_tiles_n(int N) { return ceil_div_i(N, PN); }

template <int MODE, bool BT>
int launch_gemm_panel(const GemmArgs& g, int rows, hipStream_t st) {
    if (g.N <= 0 || g.K <= 0 || rows <= 0) return MLI_ERR_BAD_ARG;
    // > 64 KiB of dynamic LDS needs the opt-in, once per kernel and device
    static std::atomic<unsigned long long> opted_in{0};
    int device = 0;
    (void)hipGetDevice(&device);
    const unsigned long long bit = 1ull << (device & 63);
    if (!(opted_in.load(std::memory_order_relaxed) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_panel_kernel<MODE, BT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelSmem);
        if (e != hipSuccess) return (int)e;
        opted_in.fetch_or(bit, std::memory_order_relaxed);
    }
    const dim3 grid(ceil_div_i(g.N, PN) * g.n_out, ceil_div_i(rows, PM), 1);
    hipLaunchKernelGGL((gemm_f32_panel_kernel<MODE, BT>), grid, dim3(kPanelThreads), kPanelSmem, st, g);
    return launch_status();
}

template int launch_gemm_panel<kPagedLatest, false>(const GemmArgs&, int, hipStream_t);
template int launch_gemm_panel<kNaiveLatest, false>(const GemmArgs&, int, hipStream_t);
template int launch_gemm_panel<kPlain, true>(const GemmArgs&, int, hipStream_t);

}  // namespace mli
