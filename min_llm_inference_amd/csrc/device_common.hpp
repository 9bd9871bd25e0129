// Shared device-side helpers for the gfx950 kernels. Wavefront width is 64 everywhere.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mli_kernels.h"

namespace mli {

constexpr int kWave = 64;
constexpr int kPage = MLI_PAGE_BLOCK_SIZE;   // tokens per page block (reference include/constants.h:12)
constexpr int kSegInp = 0;                   // reference include/constants.h:16-18
constexpr int kSegK = 1;
constexpr int kSegV = 2;

// float offset of (token slot, segment) inside one page block (reference include/utils.h:37,43)
__device__ __forceinline__ int64_t page_row_offset(int i_sequence, int emb_dim, int seg) {
    return (int64_t)(i_sequence % kPage) * emb_dim * 3 + (int64_t)seg * emb_dim;
}

// Loads through pointers that came out of LDS (page pointers) would otherwise be FLAT loads: the
// compiler cannot prove the address space.  Casting to address space 1 yields global_load_dwordx4;
// NT adds the non-temporal hint for streams that are read exactly once (K/V pages).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const f32x4_t __attribute__((address_space(1)))* gv4_ptr;

template <bool NT>
__device__ __forceinline__ float4 ldg4(const float* p) {
    f32x4_t v;
    if (NT) v = __builtin_nontemporal_load((gv4_ptr)(p));
    else v = *(gv4_ptr)(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// A pointer every lane of the wave holds the same value of (a page pointer read from LDS): move it
// to SGPRs so the loads use the scalar-base form and no per-lane 64-bit address math is needed.
__device__ __forceinline__ const float* wave_uniform(const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<const float*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// uniform base + zero-extended 32-bit per-lane byte offset: the shape the global_load saddr form wants
__device__ __forceinline__ const float* byte_offset(const float* base, unsigned bytes) {
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + bytes);
}

typedef const float __attribute__((address_space(1)))* gf_ptr;
template <bool NT>
__device__ __forceinline__ float ldg1(const float* p) {
    if (NT) return __builtin_nontemporal_load((gf_ptr)(p));
    return *(gf_ptr)(p);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

__device__ __forceinline__ void axpy4(float p, const float4& v, float4& acc) {
    acc.x = fmaf(p, v.x, acc.x);
    acc.y = fmaf(p, v.y, acc.y);
    acc.z = fmaf(p, v.z, acc.z);
    acc.w = fmaf(p, v.w, acc.w);
}

// Sum 16 per-lane partials (one per token of a page) across the 64 lanes of a wave.
// Transposing butterfly: 15 exchanges halve the live values at each of the first four
// steps, two more finish the quad.  On return every lane holds the full sum for token
// (lane >> 2) & 15.
__device__ __forceinline__ float wave_reduce16(float (&v)[16], int lane) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float send = b5 ? v[i] : v[i + 8];
        float keep = b5 ? v[i + 8] : v[i];
        a[i] = keep + __shfl_xor(send, 32, kWave);
    }
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float send = b4 ? a[i] : a[i + 4];
        float keep = b4 ? a[i + 4] : a[i];
        b[i] = keep + __shfl_xor(send, 16, kWave);
    }
    float c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float send = b3 ? b[i] : b[i + 2];
        float keep = b3 ? b[i + 2] : b[i];
        c[i] = keep + __shfl_xor(send, 8, kWave);
    }
    float send = b2 ? c[0] : c[1];
    float keep = b2 ? c[1] : c[0];
    float d = keep + __shfl_xor(send, 4, kWave);
    d += __shfl_xor(d, 2, kWave);
    d += __shfl_xor(d, 1, kWave);
    return d;
}

// ---- softmax statistics shared by the fused qkt / softmax.V kernels (see attention_scan.hip) ----------------
struct SoftmaxStats {
    float2* stats;      // [B][per_row] (max, sum exp(score - max)); nullptr = not fused
    int per_row;        // entries per row
    int chunk_tokens;   // tokens covered by one entry
};

// online (max, sum-exp) update of one wave with the scores its lanes hold (valid lanes only)
__device__ __forceinline__ void stats_accumulate(float score, bool valid, float& m, float& l) {
    const float pm = wave_max(valid ? score : -INFINITY);
    const float m_new = fmaxf(m, pm);
    const float ps = wave_sum(valid ? expf(score - m_new) : 0.f);
    l = (m == -INFINITY ? 0.f : l * expf(m - m_new)) + ps;
    m = m_new;
}

// row-level (m, l) from the row's chunk statistics: lanes read entries in parallel, two wave reductions;
// every wave of the workgroup does this redundantly (no barrier needed), every lane gets the same values
__device__ __forceinline__ void stats_merge_row(const SoftmaxStats& st, int b, int L, int lane, float& m, float& l) {
    const int n = (L + st.chunk_tokens - 1) / st.chunk_tokens;
    const float2* row = st.stats + (int64_t)b * st.per_row;
    float2 mine[4];  // up to 256 chunks per row (n_sequence <= 16384 at the smallest chunk)
    float lm = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + k * kWave;
        mine[k] = i < n ? row[i] : make_float2(-INFINITY, 0.f);
        lm = fmaxf(lm, mine[k].x);
    }
    m = wave_max(lm);
    float ls = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (mine[k].x != -INFINITY) ls += mine[k].y * expf(mine[k].x - m);
    l = wave_sum(ls);
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Caller-owned workspace = [row arrival counters of the lean single-pass scan][body: chunk statistics | partial sums].
// The counters sit at a FIXED place in front: calls of different shapes share one buffer, and a place that moved with
// the shape would sooner or later hold another call's partial sums -- the counters must stay zero between launches.
// Every extern "C" entry point converts (workspace, bytes) to the body once; the kernels' launchers see only the body.
constexpr size_t kArrivalRegionBytes = 65536;                        // 16384 rows
constexpr int kMaxArrivalRows = (int)(kArrivalRegionBytes / sizeof(unsigned));
struct WsBody {
    void* ptr;
    size_t bytes;
};
inline WsBody ws_body(void* workspace, size_t bytes) {
    if (workspace == nullptr || bytes <= kArrivalRegionBytes) return WsBody{nullptr, 0};
    return WsBody{reinterpret_cast<char*>(workspace) + kArrivalRegionBytes, bytes - kArrivalRegionBytes};
}
inline unsigned* ws_arrivals(void* body) {
    return body == nullptr ? nullptr : reinterpret_cast<unsigned*>(reinterpret_cast<char*>(body) - kArrivalRegionBytes);
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

inline int ceil_div_i(int a, int b) { return (a + b - 1) / b; }

}  // namespace mli
