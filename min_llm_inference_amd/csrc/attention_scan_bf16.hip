// bfloat16-page variants of the HBM-bound scan kernels (BASELINE config 4: bf16 KV pages, fp32 q / scores /
// accumulation / outputs).  Same decomposition as attention_scan.hip -- one wave owns whole rows, page pointers
// staged in LDS and moved to SGPRs, rows as the fast grid dimension, fixed-order combine -- with 8 elements per
// 16-byte lane load, so one load instruction covers a 512-element row (D=512 -> exactly one K or V row).
// The reference has no bf16 path; these kernels extend include/mli_kernels.h (see the header's bf16 section).
#include "device_common.hpp"

namespace mli {

constexpr int kBfThreads = 256;
constexpr int kBfWaves = kBfThreads / kWave;

int chunk_tokens_for(int n_batch, int n_sequence);  // attention_scan.hip (same heuristics / tuning knobs)
int sv_chunk_tokens_for(int n_batch, int n_sequence);
int nt_loads_for(int B, int S, int D, int esize);
int fused_softmax_wanted(int B, int S);
size_t stats_region_bytes_for(int B, int S);
int launch_softmax(float*, const int*, int, int, hipStream_t);

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef const u32x4_t __attribute__((address_space(1)))* gu4_ptr;

template <bool NT>
__device__ __forceinline__ u32x4_t ldg_u4(const void* p) {
    if (NT) return __builtin_nontemporal_load((gu4_ptr)(p));
    return *(gu4_ptr)(p);
}

__device__ __forceinline__ const void* byte_off(const void* base, unsigned bytes) {
    return reinterpret_cast<const char*>(base) + bytes;
}

__device__ __forceinline__ const uint16_t* wave_uniform16(const uint16_t* p) {
    return reinterpret_cast<const uint16_t*>(wave_uniform(reinterpret_cast<const float*>(p)));
}

// 8 bf16 in four dwords: element 2i is the low half of word i
__device__ __forceinline__ float lo_bf16(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float dot8(const float4& qa, const float4& qb, const u32x4_t& k, float acc) {
    acc = fmaf(qa.x, lo_bf16(k.x), acc);
    acc = fmaf(qa.y, hi_bf16(k.x), acc);
    acc = fmaf(qa.z, lo_bf16(k.y), acc);
    acc = fmaf(qa.w, hi_bf16(k.y), acc);
    acc = fmaf(qb.x, lo_bf16(k.z), acc);
    acc = fmaf(qb.y, hi_bf16(k.z), acc);
    acc = fmaf(qb.z, lo_bf16(k.w), acc);
    acc = fmaf(qb.w, hi_bf16(k.w), acc);
    return acc;
}

// grid = (B, ceil(S / ct)), block = 256
template <int TB, bool NT>
__global__ __launch_bounds__(kBfThreads) void qkt_paged_bf16_kernel(
    const float* __restrict__ q, const uint16_t* const* __restrict__ page_table, const int* __restrict__ lengths,
    float* __restrict__ qkt, int S, int D, int ct, SoftmaxStats st) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ float2 wave_stats[kBfWaves];
    const int b = blockIdx.x;
    const int L = lengths[b];
    const int s0 = blockIdx.y * ct;
    if (s0 >= L) return;
    const int W = S / kPage;
    const int D4 = D >> 2, D8 = D >> 3;
    const int s1 = min(s0 + ct, L);
    const int npages = (s1 - s0 + kPage - 1) / kPage;
    float4* q_sh = reinterpret_cast<float4*>(smem_raw);
    const uint16_t** ptr_sh = reinterpret_cast<const uint16_t**>(smem_raw + (size_t)D4 * 16);
    const float4* q4 = reinterpret_cast<const float4*>(q + (int64_t)b * D);
    for (int i = threadIdx.x; i < D4; i += kBfThreads) q_sh[i] = q4[i];
    for (int i = threadIdx.x; i < npages; i += kBfThreads) ptr_sh[i] = page_table[(int64_t)b * W + s0 / kPage + i];
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const float scale = sqrtf((float)D);
    const int nj = (D8 + kWave - 1) / kWave;
    const int64_t row_bytes = (int64_t)3 * D * 2;  // bytes between consecutive token slots

    // Software pipeline over half pages (8 K rows = 8 x 1 KiB per load batch): the loads of the next half page
    // are issued before the current one is consumed, so a wave always has 8-16 row loads in flight -- the
    // butterfly reduction of a page no longer leaves the wave without outstanding HBM requests.
    float run_m = -INFINITY, run_l = 0.f;
    const int pages_w = npages > wave ? (npages - wave + kBfWaves - 1) / kBfWaves : 0;
    const int n_steps = pages_w * nj;  // one step = (page of this wave, 64-lane column chunk j)
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    auto load_half = [&](u32x4_t (&buf)[8], int step, int half) {
        const int pi = wave + (step / nj) * kBfWaves;
        const int i8 = lane + (step % nj) * kWave;
        const char* krow = reinterpret_cast<const char*>(wave_uniform16(ptr_sh[pi]) + D) + half * 8 * row_bytes;
        const unsigned voff = (unsigned)i8 * 16u;
#pragma unroll
        for (int t = 0; t < 8; ++t) buf[t] = i8 < D8 ? ldg_u4<NT>(byte_off(krow + t * row_bytes, voff)) : zero4;
    };
    u32x4_t buf_a[8], buf_b[8];
    float acc[16];
    if (n_steps > 0) load_half(buf_a, 0, 0);
    for (int step = 0; step < n_steps; ++step) {
        const int j = step % nj;
        const int i8 = lane + j * kWave;
        load_half(buf_b, step, 1);
        if (j == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.f;
        }
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
        if (i8 < D8) { qa = q_sh[2 * i8]; qb = q_sh[2 * i8 + 1]; }
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = dot8(qa, qb, buf_a[t], acc[t]);
        if (step + 1 < n_steps) load_half(buf_a, step + 1, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[8 + t] = dot8(qa, qb, buf_b[t], acc[8 + t]);
        if (j == nj - 1) {
            const int pi = wave + (step / nj) * kBfWaves;
            const float tot = wave_reduce16(acc, lane);
            const int s = s0 + pi * kPage + (lane >> 2);
            const bool writer = (lane & 3) == 0 && s < L;
            const float score = tot / scale;
            if (writer) qkt[(int64_t)b * S + s] = score;
            if (st.stats != nullptr) stats_accumulate(score, writer, run_m, run_l);
        }
    }
    if (st.stats != nullptr) {
        if (lane == 0) wave_stats[wave] = make_float2(run_m, run_l);
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = -INFINITY;
            for (int w = 0; w < kBfWaves; ++w) m = fmaxf(m, wave_stats[w].x);
            float l = 0.f;
            for (int w = 0; w < kBfWaves; ++w)
                if (wave_stats[w].x != -INFINITY) l += wave_stats[w].y * expf(wave_stats[w].x - m);
            st.stats[(int64_t)b * st.per_row + blockIdx.y] = make_float2(m, l);
        }
    }
}

// grid = (B, nchunks); rows wider than one slice (64 * NJ lane-units of 8 elements) are swept slice by slice
template <int NJ, bool NT>
__global__ __launch_bounds__(kBfThreads) void softmax_v_partial_bf16_kernel(
    float* __restrict__ probs, const uint16_t* const* __restrict__ page_table, const int* __restrict__ lengths,
    float* __restrict__ dst, int S, int D, int ct, int nchunk_max, int direct, SoftmaxStats st) {
    constexpr int kSliceU = kWave * NJ;  // lane-units (of 8 elements) per slice
    constexpr int TB = 8;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* p_sh = reinterpret_cast<float*>(smem_raw);
    const uint16_t** ptr_sh = reinterpret_cast<const uint16_t**>(smem_raw + (size_t)ct * 4);
    float* red = reinterpret_cast<float*>(smem_raw + (size_t)ct * 4 + (size_t)(ct / kPage) * 8);  // [waves][slice*8]

    const int b = blockIdx.x;
    const int c = blockIdx.y;
    const int L = min(lengths[b], S);
    const int s0 = c * ct;
    const int D8 = D >> 3;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const bool fused = st.stats != nullptr;
    float* prow = probs + (int64_t)b * S + s0;
    const int span = min(ct, S - s0);

    if (s0 >= L) {
        if (fused) for (int i = threadIdx.x; i < span; i += kBfThreads) prow[i] = 0.f;
        if (direct && c == 0) {
            float* o = dst + (int64_t)b * D;
            for (int i = threadIdx.x; i < D; i += kBfThreads) o[i] = 0.f;
        }
        return;
    }
    const int s1 = min(s0 + ct, L);
    const int ntok = s1 - s0;
    const int ngroups = (ntok + kPage - 1) / kPage;
    if (fused) {
        float m, l;
        stats_merge_row(st, b, L, lane, m, l);
        const float inv_l = 1.f / l;
        for (int i = threadIdx.x; i < span; i += kBfThreads) {
            const float p = i < ntok ? expf(prow[i] - m) * inv_l : 0.f;
            prow[i] = p;
            if (i < ntok) p_sh[i] = p;
        }
    } else {
        for (int i = threadIdx.x; i < ntok; i += kBfThreads) p_sh[i] = prow[i];
    }
    for (int i = threadIdx.x; i < ngroups; i += kBfThreads)
        ptr_sh[i] = page_table[(int64_t)b * (S / kPage) + s0 / kPage + i];
    __syncthreads();

    const int64_t row_bytes = (int64_t)3 * D * 2;
    float* o = direct ? dst + (int64_t)b * D : dst + ((int64_t)b * nchunk_max + c) * D;

    auto fma8 = [](float p, const u32x4_t& v, float (&a)[8]) {
        a[0] = fmaf(p, lo_bf16(v.x), a[0]); a[1] = fmaf(p, hi_bf16(v.x), a[1]);
        a[2] = fmaf(p, lo_bf16(v.y), a[2]); a[3] = fmaf(p, hi_bf16(v.y), a[3]);
        a[4] = fmaf(p, lo_bf16(v.z), a[4]); a[5] = fmaf(p, hi_bf16(v.z), a[5]);
        a[6] = fmaf(p, lo_bf16(v.w), a[6]); a[7] = fmaf(p, hi_bf16(v.w), a[7]);
    };

    for (int u0 = 0; u0 < D8; u0 += kSliceU) {
        float acc[NJ][8];
        bool live[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
            live[j] = (u0 + lane + j * kWave) < D8;
        }
        const unsigned lane_bytes = (unsigned)(u0 + lane) * 16u;
        for (int g = wave; g < ngroups; g += kBfWaves) {
            const char* base = reinterpret_cast<const char*>(wave_uniform16(ptr_sh[g]) + 2 * (int64_t)D);  // segment 2
            const int nt = min(kPage, ntok - g * kPage);
            const float* pg = p_sh + g * kPage;
            if (nt == kPage) {
#pragma unroll 1
                for (int h = 0; h < kPage / TB; ++h) {
                    u32x4_t vb[TB][NJ];
#pragma unroll
                    for (int t = 0; t < TB; ++t)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            if (live[j]) vb[t][j] = ldg_u4<NT>(byte_off(base + (h * TB + t) * row_bytes, lane_bytes + j * kWave * 16u));
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        const float p = pg[h * TB + t];
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            if (live[j]) fma8(p, vb[t][j], acc[j]);
                    }
                }
            } else {
                for (int t = 0; t < nt; ++t) {
                    const float p = pg[t];
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if (live[j]) fma8(p, ldg_u4<NT>(byte_off(base + t * row_bytes, lane_bytes + j * kWave * 16u)), acc[j]);
                }
            }
        }
        // cross-wave sum through LDS, element-major so the final stores are contiguous floats
        if (u0 > 0) __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave * kSliceU * 8 + (lane + j * kWave) * 8 + e] = acc[j][e];
        __syncthreads();
        for (int i = threadIdx.x; i < kSliceU * 8; i += kBfThreads) {
            if (u0 * 8 + i < D) {
                float r = red[i];
#pragma unroll
                for (int w = 1; w < kBfWaves; ++w) r += red[w * kSliceU * 8 + i];
                o[u0 * 8 + i] = r;
            }
        }
    }
}

int launch_softmax_v_combine(const float* partial, const int* lengths, float* out, int B, int S, int D, int ct,
                             int nchunk, hipStream_t st);  // attention_scan.hip

static const SoftmaxStats kNoStatsBf{nullptr, 0, 0};

static int launch_qkt_paged_bf16_stats(const float* q, const uint16_t* const* page_table, const int* lengths,
                                       float* qkt, int B, int S, int D, SoftmaxStats stats, hipStream_t st) {
    if (S % kPage != 0 || D % 8 != 0 || B <= 0) return MLI_ERR_BAD_ARG;
    const int ct = chunk_tokens_for(B, S);
    const size_t smem = (size_t)D * 4 + (size_t)(ct / kPage) * 8;
    dim3 grid(B, ceil_div_i(S, ct));
    if (nt_loads_for(B, S, D, 2))
        hipLaunchKernelGGL((qkt_paged_bf16_kernel<8, true>), grid, dim3(kBfThreads), smem, st, q, page_table, lengths, qkt, S, D, ct, stats);
    else
        hipLaunchKernelGGL((qkt_paged_bf16_kernel<8, false>), grid, dim3(kBfThreads), smem, st, q, page_table, lengths, qkt, S, D, ct, stats);
    return launch_status();
}

int launch_qkt_paged_bf16(const float* q, const uint16_t* const* page_table, const int* lengths, float* qkt,
                          int B, int S, int D, hipStream_t st) {
    return launch_qkt_paged_bf16_stats(q, page_table, lengths, qkt, B, S, D, kNoStatsBf, st);
}

static int launch_softmax_v_paged_bf16_stats(float* probs, const uint16_t* const* page_table, const int* lengths,
                                             float* out, int B, int S, int D, void* workspace, size_t ws_bytes,
                                             SoftmaxStats stats, hipStream_t st) {
    if (S % kPage != 0 || D % 8 != 0 || B <= 0) return MLI_ERR_BAD_ARG;
    const int D8 = D / 8;
    const int nj = min(2, ceil_div_i(D8, kWave));
    const int slice_u = kWave * nj;
    const int ct = sv_chunk_tokens_for(B, S);
    const int nchunk = ceil_div_i(S, ct);
    const int direct = nchunk == 1;
    float* dst = out;
    if (!direct) {
        const size_t need = stats_region_bytes_for(B, S) + (size_t)B * nchunk * D * sizeof(float);
        if (workspace == nullptr || ws_bytes < need) return MLI_ERR_WORKSPACE;
        dst = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + stats_region_bytes_for(B, S));
    }
    const size_t smem = (size_t)ct * 4 + (size_t)(ct / kPage) * 8 + (size_t)kBfWaves * slice_u * 8 * 4;
    dim3 grid(B, nchunk);
    const bool nt = nt_loads_for(B, S, D, 2);
#define MLI_SVB_LAUNCH(NJ, NT)                                                                             \
    hipLaunchKernelGGL((softmax_v_partial_bf16_kernel<NJ, NT>), grid, dim3(kBfThreads), smem, st, probs, page_table, \
                       lengths, dst, S, D, ct, nchunk, direct, stats)
    if (nj == 1) { if (nt) MLI_SVB_LAUNCH(1, true); else MLI_SVB_LAUNCH(1, false); }
    else { if (nt) MLI_SVB_LAUNCH(2, true); else MLI_SVB_LAUNCH(2, false); }
#undef MLI_SVB_LAUNCH
    int rc = launch_status();
    if (rc || direct) return rc;
    return launch_softmax_v_combine(dst, lengths, out, B, S, D, ct, nchunk, st);
}

int launch_softmax_v_paged_bf16(const float* probs, const uint16_t* const* page_table, const int* lengths, float* out,
                                int B, int S, int D, void* workspace, size_t ws_bytes, hipStream_t st) {
    return launch_softmax_v_paged_bf16_stats(const_cast<float*>(probs), page_table, lengths, out, B, S, D, workspace,
                                             ws_bytes, kNoStatsBf, st);
}

int launch_scores_softmax_v_paged_bf16(const float* q, const uint16_t* const* page_table, const int* lengths,
                                       float* qkt, float* out, int B, int S, int D, void* ws, size_t ws_bytes,
                                       hipStream_t st) {
    if (!fused_softmax_wanted(B, S) || ws == nullptr || ws_bytes < stats_region_bytes_for(B, S) ||
        ceil_div_i(S, chunk_tokens_for(B, S)) > 256) {
        int rc = launch_qkt_paged_bf16(q, page_table, lengths, qkt, B, S, D, st);
        if (!rc) rc = launch_softmax(qkt, lengths, B, S, st);
        if (!rc) rc = launch_softmax_v_paged_bf16(qkt, page_table, lengths, out, B, S, D, ws, ws_bytes, st);
        return rc;
    }
    SoftmaxStats stats;
    stats.stats = reinterpret_cast<float2*>(ws);
    stats.per_row = ceil_div_i(S, 64);
    stats.chunk_tokens = chunk_tokens_for(B, S);
    int rc = launch_qkt_paged_bf16_stats(q, page_table, lengths, qkt, B, S, D, stats, st);
    if (rc) return rc;
    return launch_softmax_v_paged_bf16_stats(qkt, page_table, lengths, out, B, S, D, ws, ws_bytes, stats, st);
}

}  // namespace mli

extern "C" {

int mli_qkt_paged_bf16(const float* q_output, const mli_bf16* const* page_table, const int* lengths,
                       float* qkt_output, int n_batch, int n_sequence, int emb_dim, void* stream) {
    return mli::launch_qkt_paged_bf16(q_output, page_table, lengths, qkt_output, n_batch, n_sequence, emb_dim,
                                      mli::as_stream(stream));
}

int mli_softmax_v_paged_bf16(const float* softmax_result, const mli_bf16* const* page_table, const int* lengths,
                             float* attention_result, int n_batch, int n_sequence, int emb_dim, void* workspace,
                             size_t workspace_bytes, void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
    return mli::launch_softmax_v_paged_bf16(softmax_result, page_table, lengths, attention_result, n_batch,
                                            n_sequence, emb_dim, workspace, workspace_bytes, mli::as_stream(stream));
}

}  // extern "C"
