// HBM-bound half of the decode step: q.K^T scores, length-masked softmax, softmax.V.
// Contiguous ("naive") and paged KV layouts.  Reference behaviour being replaced:
//   src/kernels/self_attention_inference_optimized.cu:150-279 (qkt, softmax_in_place_with_lengths, softmax_v)
//   src/kernels/paged_attention.cu:208-345                    (qkt_paged_attention, softmax_v_paged_attention)
//
// Design (gfx950): one wave owns whole K/V rows -- 64 lanes x float4 = 1 KiB contiguous per
// load instruction -- so every HBM request is a full row piece; the page pointers of a
// workgroup's sequence chunk are staged once in LDS; per-row dot products are reduced with a
// transposing wave butterfly (17 exchanges per 16 tokens); the sequence is cut into fixed-size
// chunks (grid.x) so that ragged row lengths still give evenly sized work units, with a tiny
// fixed-order combine pass for softmax.V (bitwise reproducible, no float atomics).
#include <string>

#include "device_common.hpp"

namespace mli {

constexpr int kScanThreads = 256;
constexpr int kScanWaves = kScanThreads / kWave;
constexpr int kMaxChunkTokens = 1024;
constexpr int kMinChunkTokens = 64;

void set_bf16_native_mfma(int v);  // proj_gemm.hip
void set_flash_decode(int v);      // attention_fused.hip
void set_flash_variant(int v);
void set_scan_merge(int v);
void set_gemm_panel(int v);
void set_tail_tokens(int v);
void set_scan_stream(int v);
void set_scan_stream_min(int v);
void set_stream_dyn_pct(int v);
void set_stream_granule(int v);
void set_dynamic_items(int v);
void set_partial_last(int v);
void set_gemm_split(int v);
void set_row_order(int v);
void set_bf16_split(int v);
void set_prefill_fused(int v);
void set_naive_fused(int v);
void set_gemm_tall_tiles(int v);
void set_deep_k_tiles(int v);
void set_fill_compact(int v);
void set_latest_compact(int v);

// Tuning knobs (mli_tune): 0 = use the built-in heuristic / default.
static thread_local int g_chunk_tokens = 0;
static thread_local int g_nt_loads = 2;  // 0 = default cache policy, 1 = non-temporal, 2 = by working set (nt_loads_for)
static thread_local int g_qkt_token_batch = 8;

// Sequence chunk (tokens per workgroup) for the split-sequence kernels: the largest power of two in
// [64, 1024] that still yields >= min_units work units.  With rows as the fast grid dimension the choice is
// worth a few percent (measured on MI355X at B=1024, S=4096: q.K^T 256, softmax.V 512).
// MLI_CHUNK_TOKENS (power of two in [64, 1024]) overrides the heuristic for tuning runs.
constexpr int kQktUnits = 16384;  // q.K^T is insensitive to the chunk size (341-349 us for 64..512 tokens at config 4)
// softmax.V: at most 512 tokens (fewer partial sums to write and combine than q.K^T's chunks), shrunk until there are
// >= 2048 units.  Measured: B=1024, S=4096 -> 512 (8192 units); B=256, S=1024 -> 128 (26.5 us; 64: 32.8, 256: 27.8,
// 512: 44.1).
constexpr int kSvUnits = 2048;
constexpr int kSvMaxChunkTokens = 512;

static int pick_chunk_tokens(int n_batch, int n_sequence, int min_units = kQktUnits, int max_ct = kMaxChunkTokens) {
    static const int forced = [] {
        const char* e = getenv("MLI_CHUNK_TOKENS");
        const int v = e ? atoi(e) : 0;
        return (v >= kMinChunkTokens && v <= kMaxChunkTokens && (v & (v - 1)) == 0) ? v : 0;
    }();
    if (forced) return forced;
    if (g_chunk_tokens) return g_chunk_tokens;
    int ct = max_ct;
    while (ct > kMinChunkTokens && (int64_t)n_batch * ceil_div_i(n_sequence, ct) < min_units) ct >>= 1;
    return ct;
}

// ------------------------------------------------------------------------------------------
// Softmax fused into the two scan kernels (compositions only; the standalone launchers keep the reference's
// three-kernel contract).  The qkt kernels additionally emit, per (row, chunk), the chunk's running maximum and
// sum of exp(score - max); the softmax.V kernels turn the raw scores into probabilities on the fly
// (p = exp(s - m_row) / l_row, with (m_row, l_row) merged from the row's chunk statistics), write them back to
// qkt_output -- including the zero tail the reference's softmax kernel writes -- and accumulate p.V.  This
// removes the softmax launch and one read + write pass over the scores.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// qkt, paged layout.  grid = (B, ceil(S / ct)), block = 256.  Each wave takes whole pages.
// ------------------------------------------------------------------------------------------
template <int TB, bool NT>
__global__ __launch_bounds__(kScanThreads) void qkt_paged_kernel(
    const float* __restrict__ q, const float* const* __restrict__ page_table,
    const int* __restrict__ lengths, float* __restrict__ qkt, int S, int D, int ct, SoftmaxStats st) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ float2 wave_stats[kScanWaves];
    // grid = (B, chunks): the row index is the fast dimension, so consecutive workgroups (which the
    // dispatcher deals round-robin to the 8 XCDs) are different rows of the SAME chunk -- every XCD
    // gets an equal share of each chunk, and the empty high chunks are dispatched last.
    const int b = blockIdx.x;
    const int L = lengths[b];
    const int s0 = blockIdx.y * ct;
    if (s0 >= L) return;  // same early exit as the reference (paged_attention.cu:233-235)

    const int W = S / kPage;
    const int D4 = D >> 2;
    const int s1 = min(s0 + ct, L);
    const int npages = (s1 - s0 + kPage - 1) / kPage;

    float4* q_sh = reinterpret_cast<float4*>(smem_raw);                                  // D4 float4
    const float** ptr_sh = reinterpret_cast<const float**>(smem_raw + (size_t)D4 * 16);  // ct/16 pointers

    const float4* q4 = reinterpret_cast<const float4*>(q + (int64_t)b * D);
    for (int i = threadIdx.x; i < D4; i += kScanThreads) q_sh[i] = q4[i];
    for (int i = threadIdx.x; i < npages; i += kScanThreads)
        ptr_sh[i] = page_table[(int64_t)b * W + s0 / kPage + i];
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const float scale = sqrtf((float)D);  // the reference divides by sqrtf(dim), so do we
    const int nj = (D4 + kWave - 1) / kWave;

    float run_m = -INFINITY, run_l = 0.f;
    for (int pi = wave; pi < npages; pi += kScanWaves) {
        const float* page = wave_uniform(ptr_sh[pi]);
        const float* krow = page + D;  // segment 1 of token slot 0
        float acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.f;
        for (int j = 0; j < nj; ++j) {
            const int i4 = lane + j * kWave;
            if (i4 < D4) {
                const float4 qv = q_sh[i4];
                const unsigned voff = (unsigned)i4 * 16u;  // per-lane BYTE offset: one 32-bit VGPR
#pragma unroll
                for (int h = 0; h < 16 / TB; ++h) {
                    float4 kv[TB];
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        const float* trow = krow + (int64_t)(h * TB + t) * 3 * D;  // wave-uniform: stays in SGPRs
                        kv[t] = ldg4<NT>(byte_offset(trow, voff));
                    }
#pragma unroll
                    for (int t = 0; t < TB; ++t) acc[h * TB + t] = dot4(qv, kv[t], acc[h * TB + t]);
                }
            }
        }
        const float tot = wave_reduce16(acc, lane);
        const int s = s0 + pi * kPage + (lane >> 2);
        const bool writer = (lane & 3) == 0 && s < L;
        const float score = tot / scale;
        if (writer) qkt[(int64_t)b * S + s] = score;
        if (st.stats != nullptr) stats_accumulate(score, writer, run_m, run_l);
    }
    if (st.stats != nullptr) {  // merge the four waves, in wave order
        if (lane == 0) wave_stats[wave] = make_float2(run_m, run_l);
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = -INFINITY;
            for (int w = 0; w < kScanWaves; ++w) m = fmaxf(m, wave_stats[w].x);
            float l = 0.f;
            for (int w = 0; w < kScanWaves; ++w)
                if (wave_stats[w].x != -INFINITY) l += wave_stats[w].y * expf(wave_stats[w].x - m);
            st.stats[(int64_t)b * st.per_row + blockIdx.y] = make_float2(m, l);
        }
    }
}

// ------------------------------------------------------------------------------------------
// qkt, contiguous layout kt_cache[B, D, S].  Lanes run along s (float4 = 4 tokens per lane);
// the 4 waves of a workgroup split d and are summed in fixed order through LDS.
// grid = (B, ceil(S / 256)).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanThreads) void qkt_naive_kernel(
    const float* __restrict__ q, const float* __restrict__ kt, const int* __restrict__ lengths,
    float* __restrict__ qkt, int S, int D, SoftmaxStats st) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4 (*red)[kWave] = reinterpret_cast<float4 (*)[kWave]>(smem_raw);                // [waves][64] float4
    float* q_sh = reinterpret_cast<float*>(smem_raw + sizeof(float4) * kScanWaves * kWave);  // D floats
    const int b = blockIdx.x;
    const int L = lengths[b];
    const int s0 = blockIdx.y * 256;
    if (s0 >= L) return;  // reference …optimized.cu:160-162
    for (int i = threadIdx.x; i < D; i += kScanThreads) q_sh[i] = q[(int64_t)b * D + i];
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int s = s0 + lane * 4;  // S % 4 == 0, so a float4 never straddles the row end
    const bool in_row = s < S && s < L;
    const float* base = kt + (int64_t)b * D * S + s;
    const int d_per_wave = (D + kScanWaves - 1) / kScanWaves;
    const int d0 = wave * d_per_wave;
    const int d1 = min(d0 + d_per_wave, D);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_row) {
        int d = d0;
        for (; d + 8 <= d1; d += 8) {
            float4 kv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kv[u] = *reinterpret_cast<const float4*>(base + (int64_t)(d + u) * S);
#pragma unroll
            for (int u = 0; u < 8; ++u) axpy4(q_sh[d + u], kv[u], acc);
        }
        for (; d < d1; ++d) axpy4(q_sh[d], *reinterpret_cast<const float4*>(base + (int64_t)d * S), acc);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {  // whole wave stays in (the statistics below are wave-wide reductions)
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (in_row) {
            float4 r = red[0][lane];
#pragma unroll
            for (int w = 1; w < kScanWaves; ++w) {
                r.x += red[w][lane].x; r.y += red[w][lane].y; r.z += red[w][lane].z; r.w += red[w][lane].w;
            }
            const float scale = sqrtf((float)D);
            float* out = qkt + (int64_t)b * S + s;
            v[0] = r.x / scale; v[1] = r.y / scale; v[2] = r.z / scale; v[3] = r.w / scale;
            if (s + 4 <= L) {
                *reinterpret_cast<float4*>(out) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                for (int u = 0; u < 4; ++u) if (s + u < L) out[u] = v[u];   // never write past lengths[b]
            }
        }
        if (st.stats != nullptr) {
            float m = -INFINITY, l = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) stats_accumulate(v[u], in_row && s + u < L, m, l);
            if (lane == 0) st.stats[(int64_t)b * st.per_row + blockIdx.y] = make_float2(m, l);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Masked softmax in place.  One workgroup (4 waves) per row, float4 traffic; only the live prefix is read,
// the tail is zero-filled.  grid = B.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanThreads) void softmax_lengths_kernel(
    float* __restrict__ qkt, const int* __restrict__ lengths, int B, int S) {
    __shared__ float wave_part[kScanWaves];
    const int row = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int L = min(lengths[row], S);
    float4* rp = reinterpret_cast<float4*>(qkt + (int64_t)row * S);
    const int n4_len = (L + 3) >> 2;
    const int n4 = S >> 2;

    float m = -INFINITY;
    for (int i = threadIdx.x; i < n4_len; i += kScanThreads) {
        const float4 v = rp[i];
        const int s = i * 4;
        m = fmaxf(m, v.x);
        if (s + 1 < L) m = fmaxf(m, v.y);
        if (s + 2 < L) m = fmaxf(m, v.z);
        if (s + 3 < L) m = fmaxf(m, v.w);
    }
    m = wave_max(m);
    if (lane == 0) wave_part[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wave_part[0], wave_part[1]), fmaxf(wave_part[2], wave_part[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n4_len; i += kScanThreads) {
        const float4 v = rp[i];
        const int s = i * 4;
        sum += expf(v.x - m);
        if (s + 1 < L) sum += expf(v.y - m);
        if (s + 2 < L) sum += expf(v.z - m);
        if (s + 3 < L) sum += expf(v.w - m);
    }
    sum = wave_sum(sum);
    if (lane == 0) wave_part[wave] = sum;
    __syncthreads();
    sum = (wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]);
    for (int i = threadIdx.x; i < n4; i += kScanThreads) {
        const int s = i * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < L) {
            const float4 v = rp[i];
            o.x = expf(v.x - m) / sum;
            if (s + 1 < L) o.y = expf(v.y - m) / sum;
            if (s + 2 < L) o.z = expf(v.z - m) / sum;
            if (s + 3 < L) o.w = expf(v.w - m) / sum;
        }
        rp[i] = o;
    }
}

// ------------------------------------------------------------------------------------------
// softmax.V partial sums.  VEC floats per lane per load (4 when rows are 16-byte aligned),
// NJ loads per row slice; grid = (B, nchunks, d_slices).  A slice covers 64*VEC*NJ columns.
// direct != 0: single chunk per row, result goes straight to attention_result.
// ------------------------------------------------------------------------------------------
template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ void vfma(float p, const typename VecT<VEC>::type& v, typename VecT<VEC>::type& acc);
template <> __device__ __forceinline__ void vfma<4>(float p, const float4& v, float4& acc) { axpy4(p, v, acc); }
template <> __device__ __forceinline__ void vfma<1>(float p, const float& v, float& acc) { acc = fmaf(p, v, acc); }

template <int VEC> __device__ __forceinline__ typename VecT<VEC>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

template <int VEC> __device__ __forceinline__ void vadd(typename VecT<VEC>::type& a, const typename VecT<VEC>::type& b);
template <> __device__ __forceinline__ void vadd<4>(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
template <> __device__ __forceinline__ void vadd<1>(float& a, const float& b) { a += b; }

template <int VEC> __device__ __forceinline__ typename VecT<VEC>::type vload(const typename VecT<VEC>::type* p, bool nt);
template <> __device__ __forceinline__ float4 vload<4>(const float4* p, bool nt) {
    return nt ? ldg4<true>(reinterpret_cast<const float*>(p)) : ldg4<false>(reinterpret_cast<const float*>(p));
}
template <> __device__ __forceinline__ float vload<1>(const float* p, bool nt) { return nt ? ldg1<true>(p) : ldg1<false>(p); }

template <int VEC, int NJ, bool PAGED, bool NT>
__global__ __launch_bounds__(kScanThreads) void softmax_v_partial_kernel(
    float* __restrict__ probs, const void* __restrict__ src, const int* __restrict__ lengths,
    float* __restrict__ dst, int S, int D, int ct, int nchunk_max, int direct, SoftmaxStats st) {
    using V = typename VecT<VEC>::type;
    constexpr int kSliceV = kWave * NJ;          // V-elements per d-slice
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* p_sh = reinterpret_cast<float*>(smem_raw);                                   // ct floats
    const float** ptr_sh = reinterpret_cast<const float**>(smem_raw + (size_t)ct * 4);  // ct/16 pointers
    V* red = reinterpret_cast<V*>(smem_raw + (size_t)ct * 4 + (size_t)(ct / kPage) * 8);  // [waves][kSliceV]

    const int b = blockIdx.x;  // rows are the fast grid dimension (XCD balance, see qkt_paged_kernel)
    const int c = blockIdx.y;
    const int L = min(lengths[b], S);
    const int s0 = c * ct;
    const int Dv = D / VEC;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const bool fused = st.stats != nullptr;
    float* prow = probs + (int64_t)b * S + s0;
    const int span = min(ct, S - s0);            // positions of this chunk that exist in the row

    if (s0 >= L) {
        // fused: this chunk lies in the zero tail the reference's softmax kernel writes
        if (fused) for (int i = threadIdx.x; i < span; i += kScanThreads) prow[i] = 0.f;
        // an empty row still owes zeros to attention_result (reference softmax_v: result = 0)
        if (direct && c == 0) {
            V* o = reinterpret_cast<V*>(dst + (int64_t)b * D);
            for (int i = threadIdx.x; i < Dv; i += kScanThreads) o[i] = vzero<VEC>();
        }
        return;
    }
    const int s1 = min(s0 + ct, L);
    const int ntok = s1 - s0;
    if (fused) {
        float m, l;
        stats_merge_row(st, b, L, lane, m, l);
        const float inv_l = 1.f / l;
        for (int i = threadIdx.x; i < span; i += kScanThreads) {
            const float p = i < ntok ? expf(prow[i] - m) * inv_l : 0.f;
            prow[i] = p;               // probabilities (and the zero tail) replace the raw scores
            if (i < ntok) p_sh[i] = p;
        }
    } else {
        for (int i = threadIdx.x; i < ntok; i += kScanThreads) p_sh[i] = prow[i];
    }
    if (PAGED) {
        const float* const* pt = reinterpret_cast<const float* const*>(src);
        const int npages = (ntok + kPage - 1) / kPage;
        for (int i = threadIdx.x; i < npages; i += kScanThreads)
            ptr_sh[i] = pt[(int64_t)b * (S / kPage) + s0 / kPage + i];
    }
    __syncthreads();

    const int ngroups = (ntok + kPage - 1) / kPage;  // 16-token groups (pages when PAGED)
    constexpr int TB = 4;                            // rows in flight per load batch (x NJ loads each)
    const int64_t stride_f = PAGED ? 3 * (int64_t)D : (int64_t)D;  // floats between consecutive tokens
    V* o = direct ? reinterpret_cast<V*>(dst + (int64_t)b * D)
                  : reinterpret_cast<V*>(dst + ((int64_t)b * nchunk_max + c) * D);

    // rows wider than one slice (64 * NJ lane loads) are swept slice by slice; p_sh is reused
    for (int v0 = 0; v0 < Dv; v0 += kSliceV) {
        V acc[NJ];
        bool live[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc[j] = vzero<VEC>();
            live[j] = (v0 + lane + j * kWave) < Dv;
        }
        const unsigned lane_bytes = (unsigned)(v0 + lane) * (unsigned)sizeof(V);
        for (int g = wave; g < ngroups; g += kScanWaves) {
            // wave-uniform base of the group's first V row (SGPRs); lanes add one 32-bit byte offset
            const float* base = PAGED ? wave_uniform(ptr_sh[g]) + 2 * (int64_t)D
                                      : reinterpret_cast<const float*>(src) + ((int64_t)b * S + s0 + g * kPage) * D;
            const int nt = min(kPage, ntok - g * kPage);
            const float* pg = p_sh + g * kPage;
            if (nt == kPage) {
#pragma unroll 1
                for (int h = 0; h < kPage / TB; ++h) {
                    V vb[TB][NJ];
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        const float* trow = base + (int64_t)(h * TB + t) * stride_f;
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            if (live[j])
                                vb[t][j] = vload<VEC>(reinterpret_cast<const V*>(byte_offset(trow, lane_bytes + j * kWave * (unsigned)sizeof(V))), NT);
                    }
#pragma unroll
                    for (int t = 0; t < TB; ++t) {
                        const float p = pg[h * TB + t];
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            if (live[j]) vfma<VEC>(p, vb[t][j], acc[j]);
                    }
                }
            } else {
                for (int t = 0; t < nt; ++t) {
                    const float p = pg[t];
                    const float* trow = base + (int64_t)t * stride_f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if (live[j])
                            vfma<VEC>(p, vload<VEC>(reinterpret_cast<const V*>(byte_offset(trow, lane_bytes + j * kWave * (unsigned)sizeof(V))), NT), acc[j]);
                }
            }
        }
        if (v0 > 0) __syncthreads();  // the previous slice's sums have been consumed
#pragma unroll
        for (int j = 0; j < NJ; ++j) red[wave * kSliceV + lane + j * kWave] = acc[j];
        __syncthreads();
        for (int i = threadIdx.x; i < kSliceV; i += kScanThreads) {
            if (v0 + i < Dv) {
                V r = red[i];
#pragma unroll
                for (int w = 1; w < kScanWaves; ++w) vadd<VEC>(r, red[w * kSliceV + i]);
                o[v0 + i] = r;
            }
        }
    }
}

// attention_result[b, :] = sum over the row's chunks, in chunk order.  grid = (ceil(D/256), B).
__global__ __launch_bounds__(kScanThreads) void softmax_v_combine_kernel(
    const float* __restrict__ partial, const int* __restrict__ lengths, float* __restrict__ out,
    int S, int D, int ct, int nchunk_max) {
    const int b = blockIdx.y;
    const int d = blockIdx.x * kScanThreads + threadIdx.x;
    if (d >= D) return;
    const int L = min(lengths[b], S);
    const int nc = (L + ct - 1) / ct;
    const float* p = partial + (int64_t)b * nchunk_max * D + d;
    float r = 0.f;
    for (int c = 0; c < nc; ++c) r += p[(int64_t)c * D];
    out[(int64_t)b * D + d] = r;
}

__global__ __launch_bounds__(kScanThreads) void stream_copy_kernel(const float4* __restrict__ src,
                                                                   float4* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * kScanThreads + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kScanThreads;
    for (; i < n4; i += stride) dst[i] = src[i];
}

// Pure streaming read: every wave streams its own contiguous region, 16 x 1 KiB non-temporal 16-byte lane loads in
// flight, nothing written (the sums go nowhere unless they hit a value they never hit).  The read-only ceiling bench.py
// holds the scan against.
__global__ __launch_bounds__(kScanThreads) void stream_read_kernel(const float4* __restrict__ src, float* __restrict__ sink,
                                                                   size_t n4, size_t f4_per_wave) {
    constexpr int UNROLL = 16;
    const int lane = threadIdx.x & (kWave - 1);
    const size_t wave = ((size_t)blockIdx.x * kScanThreads + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * kScanThreads) >> 6;
    float acc = 0.f;
    for (size_t base = wave * f4_per_wave; base + f4_per_wave <= n4; base += nwaves * f4_per_wave) {
        for (size_t o = 0; o < f4_per_wave; o += (size_t)kWave * UNROLL) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = ldg4<true>(reinterpret_cast<const float*>(src + base + o + u * kWave + lane));
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// ------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------
int chunk_tokens_for(int n_batch, int n_sequence) { return pick_chunk_tokens(n_batch, n_sequence); }
int sv_chunk_tokens_for(int n_batch, int n_sequence) { return pick_chunk_tokens(n_batch, n_sequence, kSvUnits, kSvMaxChunkTokens); }
// K/V loads: non-temporal where the rows' K/V (upper bound B * S * D * e * 2 bytes) is far beyond the 256 MiB
// Infinity Cache -- every byte is read once per step and nothing survives to the next one --, default policy where a
// good part of it can stay on-die between two steps.  Measured (lean scan, fp32, D=256, S=1024, lengths U[S/4, 3S/4]):
// B=128 / 256: default policy 6 / 5 % faster; B=512 / 1024 / 2048: non-temporal 6 / 11 / 10 % faster.
constexpr int64_t kNtMinKvBytes = (int64_t)768 << 20;
int nt_loads_for(int B, int S, int D, int esize) {
    if (g_nt_loads != 2) return g_nt_loads;
    return (int64_t)B * S * D * esize * 2 > kNtMinKvBytes;
}
int tuned_chunk_tokens() { return (getenv("MLI_CHUNK_TOKENS") && atoi(getenv("MLI_CHUNK_TOKENS")) > 0) ? atoi(getenv("MLI_CHUNK_TOKENS")) : g_chunk_tokens; }

int launch_softmax_v_combine(const float* partial, const int* lengths, float* out, int B, int S, int D, int ct,
                             int nchunk, hipStream_t st) {
    hipLaunchKernelGGL(softmax_v_combine_kernel, dim3(ceil_div_i(D, kScanThreads), B), dim3(kScanThreads), 0, st,
                       partial, lengths, out, S, D, ct, nchunk);
    return launch_status();
}

// Workspace layout (mli_attention_workspace_bytes): [row arrival counters: kArrivalRegionBytes, device_common.hpp] then the
// body every launcher here sees: [chunk statistics: B * ceil(S/64) float2, 256-B aligned][partial sums: B * nchunk * D floats].
static size_t stats_region_bytes(int B, int S) {
    const size_t n = (size_t)B * ceil_div_i(S, kMinChunkTokens) * sizeof(float2);
    return (n + 255) & ~(size_t)255;
}

static SoftmaxStats stats_view(void* workspace, int B, int S, int chunk_tokens) {
    SoftmaxStats st;
    st.stats = reinterpret_cast<float2*>(workspace);
    st.per_row = ceil_div_i(S, kMinChunkTokens);
    st.chunk_tokens = chunk_tokens;
    return st;
}

template <int VEC, bool PAGED>
static int launch_softmax_v_impl(float* probs, const void* src, const int* lengths, float* out,
                                 int B, int S, int D, void* workspace, size_t ws_bytes, SoftmaxStats st_in,
                                 hipStream_t st) {
    const int Dv = D / VEC;
    const int nj = min(2, ceil_div_i(Dv, kWave));  // <= 64 VGPRs -> 8 waves/SIMD; wider rows are swept in slices
    const int slice_v = kWave * nj;
    int ct = pick_chunk_tokens(B, S, kSvUnits, kSvMaxChunkTokens);
    const int nchunk = ceil_div_i(S, ct);
    const int direct = nchunk == 1;
    float* dst = out;
    if (!direct) {
        const size_t need = stats_region_bytes(B, S) + (size_t)B * nchunk * D * sizeof(float);
        if (workspace == nullptr || ws_bytes < need) return MLI_ERR_WORKSPACE;
        dst = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + stats_region_bytes(B, S));
    }
    const size_t smem = (size_t)ct * 4 + (size_t)(ct / kPage) * 8 + (size_t)kScanWaves * slice_v * VEC * 4;
    dim3 grid(B, nchunk);
#define MLI_SV_LAUNCH(NJ)                                                                                  \
    do {                                                                                                   \
        if (nt_loads_for(B, S, D, 4))                                                                      \
            hipLaunchKernelGGL((softmax_v_partial_kernel<VEC, NJ, PAGED, true>), grid, dim3(kScanThreads), smem, st, \
                               probs, src, lengths, dst, S, D, ct, nchunk, direct, st_in);                 \
        else                                                                                               \
            hipLaunchKernelGGL((softmax_v_partial_kernel<VEC, NJ, PAGED, false>), grid, dim3(kScanThreads), smem, st, \
                               probs, src, lengths, dst, S, D, ct, nchunk, direct, st_in);                 \
    } while (0)
    if (nj == 1) MLI_SV_LAUNCH(1);
    else MLI_SV_LAUNCH(2);
#undef MLI_SV_LAUNCH
    int rc = launch_status();
    if (rc || direct) return rc;
    return launch_softmax_v_combine(dst, lengths, out, B, S, D, ct, nchunk, st);
}

static const SoftmaxStats kNoStats{nullptr, 0, 0};

int launch_qkt_paged_stats(const float* q, const float* const* page_table, const int* lengths, float* qkt,
                           int B, int S, int D, SoftmaxStats stats, hipStream_t st) {
    if (S % kPage != 0 || D % 4 != 0 || B <= 0) return MLI_ERR_BAD_ARG;
    const int ct = pick_chunk_tokens(B, S);
    const size_t smem = (size_t)D * 4 + (size_t)(ct / kPage) * 8;
    dim3 grid(B, ceil_div_i(S, ct));
#define MLI_QKT_LAUNCH(TB, NT) \
    hipLaunchKernelGGL((qkt_paged_kernel<TB, NT>), grid, dim3(kScanThreads), smem, st, q, page_table, lengths, qkt, S, D, ct, stats)
    const bool nt_kv = nt_loads_for(B, S, D, 4);
    if (g_qkt_token_batch == 16) { if (nt_kv) MLI_QKT_LAUNCH(16, true); else MLI_QKT_LAUNCH(16, false); }
    else if (g_qkt_token_batch == 4) { if (nt_kv) MLI_QKT_LAUNCH(4, true); else MLI_QKT_LAUNCH(4, false); }
    else { if (nt_kv) MLI_QKT_LAUNCH(8, true); else MLI_QKT_LAUNCH(8, false); }
#undef MLI_QKT_LAUNCH
    return launch_status();
}

int launch_qkt_paged(const float* q, const float* const* page_table, const int* lengths, float* qkt,
                     int B, int S, int D, hipStream_t st) {
    return launch_qkt_paged_stats(q, page_table, lengths, qkt, B, S, D, kNoStats, st);
}

int launch_qkt_naive_stats(const float* q, const float* kt, const int* lengths, float* qkt, int B, int S, int D,
                           SoftmaxStats stats, hipStream_t st) {
    if (S % 4 != 0 || B <= 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL(qkt_naive_kernel, dim3(B, ceil_div_i(S, 256)), dim3(kScanThreads),
                       sizeof(float4) * kScanWaves * kWave + (size_t)D * 4, st, q, kt, lengths, qkt, S, D, stats);
    return launch_status();
}

int launch_qkt_naive(const float* q, const float* kt, const int* lengths, float* qkt, int B, int S, int D,
                     hipStream_t st) {
    return launch_qkt_naive_stats(q, kt, lengths, qkt, B, S, D, kNoStats, st);
}

int launch_softmax(float* qkt, const int* lengths, int B, int S, hipStream_t st) {
    if (S % 4 != 0 || B <= 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL(softmax_lengths_kernel, dim3(B), dim3(kScanThreads), 0, st, qkt, lengths, B, S);
    return launch_status();
}

int launch_softmax_v_naive(const float* probs, const float* v_cache, const int* lengths, float* out,
                           int B, int S, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    if (B <= 0 || S <= 0 || D <= 0) return MLI_ERR_BAD_ARG;
    float* p = const_cast<float*>(probs);  // only written in the fused mode
    if (D % 4 == 0) return launch_softmax_v_impl<4, false>(p, v_cache, lengths, out, B, S, D, ws, ws_bytes, kNoStats, st);
    return launch_softmax_v_impl<1, false>(p, v_cache, lengths, out, B, S, D, ws, ws_bytes, kNoStats, st);
}

int launch_softmax_v_paged(const float* probs, const float* const* page_table, const int* lengths, float* out,
                           int B, int S, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    if (S % kPage != 0 || D % 4 != 0 || B <= 0) return MLI_ERR_BAD_ARG;
    return launch_softmax_v_impl<4, true>(const_cast<float*>(probs), page_table, lengths, out, B, S, D, ws, ws_bytes,
                                          kNoStats, st);
}

// ---- fused compositions: scores + chunk statistics, then softmax folded into softmax.V ------------------
// 1 = always fuse, 0 = never, -1 (default) = fuse when the step is launch-bound.  Measured on MI355X: folding the
// softmax into softmax.V lengthens every workgroup's prologue (statistics -> exp -> write-back before the first V
// load), which costs ~10 us at B*S = 4M (BASELINE config 4) but saves a launch and a pass over the scores, worth
// 10-15 % of the step at B*S = 256K (configs 2/3).
static thread_local int g_fused_softmax = -1;

static bool can_fuse(int B, int S, int stats_chunk, void* ws, size_t ws_bytes) {
    const bool want = g_fused_softmax == 1 || (g_fused_softmax < 0 && (int64_t)B * S <= (1 << 20));
    // stats_merge_row() holds at most 256 chunk entries per row in registers
    return want && ws != nullptr && ws_bytes >= stats_region_bytes(B, S) && ceil_div_i(S, stats_chunk) <= 256;
}

int launch_scores_softmax_v_paged(const float* q, const float* const* page_table, const int* lengths, float* qkt,
                                  float* out, int B, int S, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!can_fuse(B, S, pick_chunk_tokens(B, S), ws, ws_bytes)) {
        int rc = launch_qkt_paged(q, page_table, lengths, qkt, B, S, D, st);
        if (!rc) rc = launch_softmax(qkt, lengths, B, S, st);
        if (!rc) rc = launch_softmax_v_paged(qkt, page_table, lengths, out, B, S, D, ws, ws_bytes, st);
        return rc;
    }
    const SoftmaxStats stats = stats_view(ws, B, S, pick_chunk_tokens(B, S));
    int rc = launch_qkt_paged_stats(q, page_table, lengths, qkt, B, S, D, stats, st);
    if (rc) return rc;
    return launch_softmax_v_impl<4, true>(qkt, page_table, lengths, out, B, S, D, ws, ws_bytes, stats, st);
}

int launch_scores_softmax_v_naive(const float* q, const float* kt, const float* v_cache, const int* lengths,
                                  float* qkt, float* out, int B, int S, int D, void* ws, size_t ws_bytes,
                                  hipStream_t st) {
    if (!can_fuse(B, S, 256, ws, ws_bytes)) {
        int rc = launch_qkt_naive(q, kt, lengths, qkt, B, S, D, st);
        if (!rc) rc = launch_softmax(qkt, lengths, B, S, st);
        if (!rc) rc = launch_softmax_v_naive(qkt, v_cache, lengths, out, B, S, D, ws, ws_bytes, st);
        return rc;
    }
    const SoftmaxStats stats = stats_view(ws, B, S, 256);  // qkt_naive_kernel covers 256 tokens per workgroup
    int rc = launch_qkt_naive_stats(q, kt, lengths, qkt, B, S, D, stats, st);
    if (rc) return rc;
    if (D % 4 == 0) return launch_softmax_v_impl<4, false>(qkt, v_cache, lengths, out, B, S, D, ws, ws_bytes, stats, st);
    return launch_softmax_v_impl<1, false>(qkt, v_cache, lengths, out, B, S, D, ws, ws_bytes, stats, st);
}

// exported to attention_scan_bf16.hip
size_t stats_region_bytes_for(int B, int S) { return stats_region_bytes(B, S); }
static size_t partial_region_bytes(int B, int S, int D) {
    const size_t nchunk = (size_t)ceil_div_i(S, kMinChunkTokens);
    return nchunk <= 1 ? 0 : (size_t)B * nchunk * (size_t)D * sizeof(float);
}
int fused_softmax_wanted(int B, int S) {
    return g_fused_softmax == 1 || (g_fused_softmax < 0 && (int64_t)B * S <= (1 << 20));
}

}  // namespace mli

extern "C" {

size_t mli_attention_workspace_bytes(int n_batch, int n_sequence, int dim) {
    if (n_batch <= 0 || n_sequence <= 0 || dim <= 0) return 0;
    // [row arrival counters, fixed size][chunk statistics][partial sums], the last two sized for the smallest chunk the
    // heuristic (or the tuning knob) may choose
    return mli::kArrivalRegionBytes + mli::stats_region_bytes(n_batch, n_sequence) +
           mli::partial_region_bytes(n_batch, n_sequence, dim);
}

int mli_attention_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace == nullptr || workspace_bytes < mli::kArrivalRegionBytes) return MLI_ERR_WORKSPACE;
    return (int)hipMemsetAsync(workspace, 0, mli::kArrivalRegionBytes, mli::as_stream(stream));
}

int mli_qkt(const float* q_output, const float* kt_cache, const int* lengths, float* qkt_output,
            int n_batch, int n_sequence, int dim, void* stream) {
    return mli::launch_qkt_naive(q_output, kt_cache, lengths, qkt_output, n_batch, n_sequence, dim,
                                 mli::as_stream(stream));
}

int mli_softmax_in_place_with_lengths(float* qkt_output, const int* lengths, int n_batch, int n_sequence,
                                      void* stream) {
    return mli::launch_softmax(qkt_output, lengths, n_batch, n_sequence, mli::as_stream(stream));
}

int mli_softmax_v(const float* softmax_result, const float* v_cache, const int* lengths, float* attention_result,
                  int n_batch, int n_sequence, int output_dim, void* workspace, size_t workspace_bytes,
                  void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
    return mli::launch_softmax_v_naive(softmax_result, v_cache, lengths, attention_result, n_batch, n_sequence,
                                       output_dim, workspace, workspace_bytes, mli::as_stream(stream));
}

int mli_qkt_paged(const float* q_output, const float* const* page_table, const int* lengths, float* qkt_output,
                  int n_batch, int n_sequence, int emb_dim, void* stream) {
    return mli::launch_qkt_paged(q_output, page_table, lengths, qkt_output, n_batch, n_sequence, emb_dim,
                                 mli::as_stream(stream));
}

int mli_softmax_v_paged(const float* softmax_result, const float* const* page_table, const int* lengths,
                        float* attention_result, int n_batch, int n_sequence, int emb_dim, void* workspace,
                        size_t workspace_bytes, void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
    return mli::launch_softmax_v_paged(softmax_result, page_table, lengths, attention_result, n_batch, n_sequence,
                                       emb_dim, workspace, workspace_bytes, mli::as_stream(stream));
}

int mli_tune(const char* key, int value) {
    const std::string k(key ? key : "");
    if (k == "chunk_tokens") {
        if (value != 0 && (value < mli::kMinChunkTokens || value > mli::kMaxChunkTokens || (value & (value - 1)))) return MLI_ERR_BAD_ARG;
        mli::g_chunk_tokens = value;
    } else if (k == "nt_loads") {
        mli::g_nt_loads = value < 0 || value > 2 ? 2 : value;
    } else if (k == "latest_compact") {
        mli::set_latest_compact(value);
    } else if (k == "fill_compact") {
        mli::set_fill_compact(value);
    } else if (k == "gemm_panel") {
        mli::set_gemm_panel(value);
    } else if (k == "gemm_deep_k") {
        mli::set_deep_k_tiles(value);
    } else if (k == "gemm_tall_tiles") {
        mli::set_gemm_tall_tiles(value);
    } else if (k == "naive_scan_fused") {
        mli::set_naive_fused(value);
    } else if (k == "gemm_bf16_split") {
        mli::set_bf16_split(value);
    } else if (k == "prefill_fused") {
        mli::set_prefill_fused(value);
    } else if (k == "scan_row_order") {
        mli::set_row_order(value);
    } else if (k == "gemm_split") {
        mli::set_gemm_split(value);
    } else if (k == "scan_partial_last") {
        mli::set_partial_last(value);
    } else if (k == "scan_stream") {
        mli::set_scan_stream(value);
    } else if (k == "scan_stream_dynamic_pct") {
        mli::set_stream_dyn_pct(value);
    } else if (k == "scan_stream_granule") {
        mli::set_stream_granule(value);
    } else if (k == "scan_stream_min_tokens") {
        mli::set_scan_stream_min(value);
    } else if (k == "scan_tail_tokens") {
        if (value != 0 && (value < 64 || value > 1024 || (value & (value - 1)))) return MLI_ERR_BAD_ARG;
        mli::set_tail_tokens(value);
    } else if (k == "scan_dynamic_items") {
        mli::set_dynamic_items(value);
    } else if (k == "scan_merge") {
        mli::set_scan_merge(value);
    } else if (k == "flash_variant") {
        mli::set_flash_variant(value);
    } else if (k == "flash_decode") {
        mli::set_flash_decode(value);
    } else if (k == "fused_softmax") {
        mli::g_fused_softmax = value < 0 ? -1 : (value != 0);
    } else if (k == "bf16_native_mfma") {
        mli::set_bf16_native_mfma(value);
    } else if (k == "qkt_token_batch") {
        if (value != 4 && value != 8 && value != 16) return MLI_ERR_BAD_ARG;
        mli::g_qkt_token_batch = value;
    } else {
        return MLI_ERR_BAD_ARG;
    }
    return 0;
}

int mli_stream_copy(const float* src, float* dst, size_t n_floats, void* stream) {
    if (n_floats % 4 != 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL(mli::stream_copy_kernel, dim3(256 * 16), dim3(mli::kScanThreads), 0, mli::as_stream(stream),
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n_floats / 4);
    return mli::launch_status();
}

int mli_stream_read(const float* src, float* sink, size_t n_floats, void* stream) {
    const size_t f4_per_wave = 4096;   // 64 KiB per wave and pass
    if (n_floats % (4 * f4_per_wave) != 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL(mli::stream_read_kernel, dim3(256 * 8), dim3(mli::kScanThreads), 0, mli::as_stream(stream),
                       reinterpret_cast<const float4*>(src), sink, n_floats / 4, f4_per_wave);
    return mli::launch_status();
}

}  // extern "C"
