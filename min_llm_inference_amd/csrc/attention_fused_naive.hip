// Single-launch decode attention over the CONTIGUOUS caches (kt_cache[B, D, S] -- K transposed --, v_cache[B, S, D]):
// the lean form of the reference's inference_self_attention (src/kernels/self_attention_inference_optimized.cu:282-301
// = launch_qkt + launch_softmax_in_place_with_lengths + launch_softmax_v) for callers that do not read qkt_output.
//
// K is stored transposed, so a token's K is not a row that could be consumed next to its V row as in the paged scan;
// what CAN be fused is the chunk: one workgroup takes 256 tokens of one row and
//   1. scores them from the K^T tile [D][256] -- lanes along the tokens (float4 = 4 tokens per lane, 1 KiB per wave
//      load), the 4 waves split d and are summed in wave order through LDS (the layout and order of qkt_naive_kernel);
//   2. turns the scores into un-normalised probabilities exp(score - chunk max) in LDS, never in memory;
//   3. accumulates them over the V tile [256][D] -- lanes along d, the waves split the tokens, summed in wave order;
//   4. publishes (chunk max, chunk sum, partial output row); the workgroup whose arrival completes the row merges the
//      row's chunks in chunk order (row_publish_merge: the paged scan's hand-off).
// Against the three-launch composition this drops two launches, the raw-score / probability round trip and the softmax
// kernel's pass; the K^T and V bytes are read once each either way.  BASELINE config 2 (B=256, D=256, S=1024):
// qkt 27 us + softmax_v 27 us + combine 5 us -> one launch.
// Results differ from the materialising composition by fp32 rounding of the merge only (tested: <= 1e-5).
#include "scan_item_body.hpp"

namespace mli {

size_t stats_region_bytes_for(int B, int S);             // attention_scan.hip
int nt_loads_for(int B, int S, int D, int esize);

constexpr int kNvThreads = 256;
constexpr int kNvWaves = kNvThreads / kWave;
constexpr int kNvChunk = 256;  // tokens per workgroup: 64 lanes x float4 along s
constexpr int kNvRows = 8;     // V rows in flight per wave and load batch

// grid = (B, ceil(S / 256)), rows fast.  NJ = float4 lane loads per V row and wave (row width <= NJ * 256 floats per
// sweep; wider rows are swept in slices).
template <int NJ, bool NT>
__global__ __launch_bounds__(kNvThreads) void naive_decode_scan_kernel(
    const float* __restrict__ q, const float* __restrict__ kt, const float* __restrict__ v,
    const int* __restrict__ lengths, float* __restrict__ out, float2* ml, float* partial, int S, int D, int ml_per_row,
    int nchunk, int direct, unsigned* arrivals) {
    constexpr int kSlice4 = kWave * NJ;  // float4 units of one V sweep
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4* red = reinterpret_cast<float4*>(smem_raw);                       // [waves][64]   score partials
    float4* ored = red + kNvWaves * kWave;                                  // [waves][kSlice4] output partials
    float* p_sh = reinterpret_cast<float*>(ored + kNvWaves * kSlice4);      // [256] exp(score - chunk max)
    float* q_sh = p_sh + kNvChunk;                                          // [D]
    __shared__ float2 chunk_ml;
    __shared__ int last_sh;

    const int b = blockIdx.x, c = blockIdx.y;
    const int L = min(max(lengths[b], 0), S);
    const int s0 = c * kNvChunk;
    const int D4 = D >> 2;
    if (L == 0) {  // empty slot: zeros (reference softmax_v: result = 0), written once
        if (c == 0) for (int i = threadIdx.x; i < D; i += kNvThreads) out[(int64_t)b * D + i] = 0.f;
        return;
    }
    if (s0 >= L) return;
    for (int i = threadIdx.x; i < D; i += kNvThreads) q_sh[i] = q[(int64_t)b * D + i];
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;

    const int ntok = min(kNvChunk, L - s0);
    const int ngroups = (ntok + 15) >> 4;
    // V batches of this wave, in order: (group g = wave, wave + 4, ...; rows 0..7, then 8..15 of the group)
    const int n_vbatch = ngroups > wave ? 2 * ((ngroups - wave + kNvWaves - 1) / kNvWaves) : 0;
    auto load_v = [&](int i, int v0, float4 (&r)[kNvRows][NJ]) {
        const int g = wave + kNvWaves * (i >> 1), h = (i & 1) * kNvRows;
        const int nt = min(16, ntok - g * 16);
        const float* base = v + ((int64_t)b * S + s0 + g * 16 + h) * D + (int64_t)(v0 + lane) * 4;
#pragma unroll
        for (int t = 0; t < kNvRows; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                r[t][j] = (i < n_vbatch && h + t < nt && v0 + lane + j * kWave < D4)
                              ? ldg4<NT>(base + (int64_t)t * D + j * kWave * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 va[kNvRows][NJ], vb[kNvRows][NJ];

    // ---- 1. scores of tokens s0 + 4 * lane .. + 3: this wave's quarter of d; loads run one batch of 8 rows ahead ----
    {
        const int s = s0 + lane * 4;  // S % 4 == 0: a float4 never straddles the row end
        const bool in_row = s < L;
        const float* base = kt + (int64_t)b * D * S + s;
        const int d_per_wave = (D + kNvWaves - 1) / kNvWaves;
        const int d0 = wave * d_per_wave;
        const int d1 = min(d0 + d_per_wave, D);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        auto load_k = [&](int d, float4 (&r)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                r[u] = (in_row && d + u < d1) ? ldg4<NT>(base + (int64_t)(d + u) * S) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto mac_k = [&](int d, const float4 (&r)[8]) {  // d ascending: the order of qkt_naive_kernel
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (d + u < d1) axpy4(q_sh[d + u], r[u], acc);
        };
        float4 ka[8], kb[8];
        load_k(d0, ka);
        for (int d = d0; d < d1; d += 16) {
            load_k(d + 8, kb);
            mac_k(d, ka);
            load_k(d + 16, ka);
            mac_k(d + 8, kb);
        }
        load_v(0, 0, va);  // the first V batch travels while the scores are reduced
        red[wave * kWave + lane] = acc;
        __syncthreads();
        if (wave == 0) {  // the whole wave stays in: the statistics are wave-wide reductions
            float4 r = red[lane];
#pragma unroll
            for (int w = 1; w < kNvWaves; ++w) {
                const float4 o = red[w * kWave + lane];
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            const float scale = sqrtf((float)D);
            const float sc[4] = {r.x / scale, r.y / scale, r.z / scale, r.w / scale};
            float lm = -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u) lm = (in_row && s + u < L) ? fmaxf(lm, sc[u]) : lm;
            const float m = wave_max(lm);  // finite: token s0 is live
            float ps = 0.f;
            float pr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pr[u] = (in_row && s + u < L) ? expf(sc[u] - m) : 0.f;
                ps += pr[u];
            }
            const float l = wave_sum(ps);
            *reinterpret_cast<float4*>(p_sh + lane * 4) = make_float4(pr[0], pr[1], pr[2], pr[3]);
            if (lane == 0) chunk_ml = make_float2(m, l);
        }
        __syncthreads();
    }
    const float m = chunk_ml.x, l = chunk_ml.y;

    // ---- 2. partial output: sum over the chunk's tokens of p . V, groups of 16 tokens dealt to the waves ----
    const float norm = direct ? 1.f / l : 1.f;
    float* o_row = direct ? out + (int64_t)b * D : partial + ((int64_t)b * nchunk + c) * D;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(o_row, 0, D * (int)sizeof(float), 0x00020000);
    for (int v0 = 0; v0 < D4; v0 += kSlice4) {
        float4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto mac_v = [&](int i, const float4 (&r)[kNvRows][NJ]) {  // tokens ascending within the wave's groups
            const int g = wave + kNvWaves * (i >> 1), h = (i & 1) * kNvRows;
            const int nt = min(16, ntok - g * 16);
            const float* pg = p_sh + g * 16 + h;
#pragma unroll
            for (int t = 0; t < kNvRows; ++t) {
                if (i < n_vbatch && h + t < nt) {  // wave-uniform: never multiply memory beyond the row's length
                    const float p = pg[t];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) axpy4(p, r[t][j], acc[j]);
                }
            }
        };
        if (v0 > 0) load_v(0, v0, va);
        for (int i = 0; i < n_vbatch; i += 2) {
            load_v(i + 1, v0, vb);
            mac_v(i, va);
            load_v(i + 2, v0, va);
            mac_v(i + 1, vb);
        }
        if (v0 > 0) __syncthreads();  // the previous slice's sums have been consumed
#pragma unroll
        for (int j = 0; j < NJ; ++j) ored[wave * kSlice4 + lane + j * kWave] = acc[j];
        __syncthreads();
        for (int i = threadIdx.x; i < kSlice4; i += kNvThreads) {
            if (v0 + i < D4) {
                float4 r = ored[i];
#pragma unroll
                for (int w = 1; w < kNvWaves; ++w) {
                    const float4 o = ored[w * kSlice4 + i];
                    r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
                }
                if (direct) {
                    *reinterpret_cast<float4*>(o_row + (int64_t)(v0 + i) * 4) = make_float4(r.x * norm, r.y * norm, r.z * norm, r.w * norm);
                } else {  // write-through: another workgroup of this launch reads the row back
                    fu_u32x4 raw;
                    raw.x = __float_as_uint(r.x); raw.y = __float_as_uint(r.y); raw.z = __float_as_uint(r.z); raw.w = __float_as_uint(r.w);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, orsrc, (v0 + i) * 16, 0, 16);
                }
            }
        }
    }
    if (direct) return;
    __syncthreads();  // every read of the LDS scratch the merge reuses is done
    row_publish_merge<kNvThreads>(m, l, ml + (int64_t)b * ml_per_row, c, (L + kNvChunk - 1) / kNvChunk, arrivals + b,
                                  partial + (int64_t)b * nchunk * D, D, out + (int64_t)b * D, reinterpret_cast<float*>(red),
                                  &last_sh);
}

static thread_local int g_naive_fused = 1;  // mli_tune "naive_scan_fused": 0 = the lean contiguous composition is not offered
void set_naive_fused(int v) { g_naive_fused = v != 0; }

static inline bool aligned16_nv(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// 1 = ran, 0 = shape not covered (the caller takes the three-launch composition), else an error (+1 if positive).
// ws = workspace BODY (the arrival counters sit in front of it).
int launch_fused_decode_naive(const float* q, const float* kt, const float* v, const int* lengths, float* out, int B,
                              int S, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!g_naive_fused || B <= 0 || D % 4 != 0 || S % 4 != 0 || D <= 0 || S <= 0) return 0;
    if (!aligned16_nv(kt) || !aligned16_nv(v) || !aligned16_nv(out)) return 0;
    const int nchunk = ceil_div_i(S, kNvChunk);
    // the last arriver keeps the row's chunk statistics in the 4 KiB score-partial buffer: 512 (max, sum) pairs
    if (nchunk > kNvWaves * kWave * 2) return 0;
    const int direct = nchunk == 1;
    const int ml_per_row = ceil_div_i(S, 64);
    float2* ml = nullptr;
    float* partial = nullptr;
    unsigned* arrivals = nullptr;
    if (!direct) {
        const size_t stats_bytes = stats_region_bytes_for(B, S);
        if (ws == nullptr || ws_bytes < stats_bytes + (size_t)B * nchunk * D * sizeof(float) || B > kMaxArrivalRows / 2) return 0;
        ml = reinterpret_cast<float2*>(ws);
        partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + stats_bytes);
        arrivals = ws_arrivals(ws);
    }
    const int nj = (D >> 2) <= kWave ? 1 : 2;
    const size_t smem = sizeof(float4) * kNvWaves * kWave + sizeof(float4) * kNvWaves * kWave * nj + sizeof(float) * kNvChunk +
                        sizeof(float) * (size_t)D;
    if (smem > 64 * 1024) return 0;
    const dim3 grid(B, nchunk);
    const bool nt = nt_loads_for(B, S, D, 4);
#define MLI_NV_LAUNCH(NJ, NT)                                                                                          \
    hipLaunchKernelGGL((naive_decode_scan_kernel<NJ, NT>), grid, dim3(kNvThreads), smem, st, q, kt, v, lengths, out, ml, \
                       partial, S, D, ml_per_row, nchunk, direct, arrivals)
    if (nj == 1) {
        if (nt) MLI_NV_LAUNCH(1, true);
        else MLI_NV_LAUNCH(1, false);
    } else {
        if (nt) MLI_NV_LAUNCH(2, true);
        else MLI_NV_LAUNCH(2, false);
    }
#undef MLI_NV_LAUNCH
    const int rc = launch_status();
    return rc ? (rc > 0 ? rc + 1 : rc) : 1;
}

}  // namespace mli

// the scan alone (what mli_self_attention_lean runs after the projection), for hosts that do their own projection and
// for bench.py's per-kernel timing
extern "C" int mli_decode_scan_contiguous(const float* q_output, const float* kt_cache, const float* v_cache,
                                          const int* lengths, float* attention_result, int n_batch, int n_sequence,
                                          int emb_dim, void* workspace, size_t workspace_bytes, void* stream) {
    const mli::WsBody body = mli::ws_body(workspace, workspace_bytes);
    const int r = mli::launch_fused_decode_naive(q_output, kt_cache, v_cache, lengths, attention_result, n_batch, n_sequence,
                                                 emb_dim, body.ptr, body.bytes, mli::as_stream(stream));
    if (r == 1) return 0;
    if (r == 0) return MLI_ERR_BAD_ARG;
    return r < 0 ? r : r - 1;
}
