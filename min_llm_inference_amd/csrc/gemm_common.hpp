// Shared by the fp32 (proj_gemm.hip) and bf16 (proj_gemm_bf16.hip) MFMA GEMM kernels: the argument block and
// the per-row source / destination resolution (page-table lookups happen here, once per workgroup).
#pragma once

#include <type_traits>

#include "device_common.hpp"

namespace mli {

enum GemmMode : int {
    kNaiveLatest = 0,
    kNaiveFill = 1,
    kPagedLatest = 2,
    kPagedFill = 3,
    kPlain = 4,  // C[M,N] = A[M,K] . B  (B is [K,N], or [N,K] when b_transposed)
};

struct GemmArgs {
    // weights / B operands: up to three [K, N] matrices (k, q, v) -- or one [N, K] matrix (plain, transposed)
    const float* w[3];
    int n_out;       // how many of w[] are live
    int out_id[3];   // which output each live weight feeds: 0 = K, 1 = Q, 2 = V
    int M, N, K;     // M = rows per z-slice upper bound, N = out dim, K = in dim
    // row sources / sinks
    const float* a_plain;  // kPlain: A
    float* c_plain;        // kPlain: C
    int lda, ldc;
    const float* inp_embedding;  // naive: [B, S, K]
    float* kt_cache;             // naive: [B, N, S]
    float* v_cache;              // naive: [B, S, N]
    float* const* page_table;    // paged: [B, S/16]
    float* q_output;             // latest: [B, N]
    const int* lengths;
    const int* new_batch_idx;    // fill
    int B, S;
    int n_new;       // fill: number of live entries of new_batch_idx
    int compact;     // 1 = the M dimension is a flat list built on the device (see FillIndex): all (new row, token)
                     // pairs for the fill modes, the non-empty batch rows for the latest modes
    // kPlain only: when set, C is NOT stored; every workgroup reduces its tile to one (max, lowest index of the max)
    // pair per row and writes it to row_best[m * tiles_n + tile] -- the decoder's argmax as the logits GEMM's epilogue
    struct RowBest* row_best;
    // fill modes only: when emb_table is set, the A rows are not READ from the input-embedding segment but computed
    // on the fly as emb_table[inp[b, s]] + wpe[s] -- the encoder as the GEMM's prologue -- and the
    // workgroups of the first column tile also WRITE them there (the decode projection reads position L - 1 later)
    const float* emb_table;  // [n_vocab, K]
    const float* wpe;        // [S, K]
    const int* inp;          // [B, S] token ids
};

struct RowBest {
    float value;
    int index;
};

// argmax order: larger value wins, equal values -> the lower index (the reference's host decoder,
// tests/test_utils.cpp:607-614); index -1 = "nothing yet" compares as the highest index
__device__ __forceinline__ void argmax_take(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && (unsigned)oi < (unsigned)i)) {
        v = ov;
        i = oi;
    }
}

// One step of the argmax butterfly: of its 2 * H candidate rows a lane keeps H (the upper ones when `upper`) and
// merges into them what its partner (lane ^ mask) held for the same rows.
template <int H>
__device__ __forceinline__ void argmax_butterfly_step(float (&v)[16], int (&ix)[16], bool upper, int mask) {
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const float send_v = upper ? v[k] : v[k + H];
        const int send_i = upper ? ix[k] : ix[k + H];
        float keep_v = upper ? v[k + H] : v[k];
        int keep_i = upper ? ix[k + H] : ix[k];
        const float ov = __shfl_xor(send_v, mask, kWave);
        const int oi = __shfl_xor(send_i, mask, kWave);
        const bool take = ov > keep_v || (ov == keep_v && (unsigned)oi < (unsigned)keep_i);
        v[k] = take ? ov : keep_v;
        ix[k] = take ? oi : keep_i;
    }
}

// Prefill over a FLAT row list.  A new row's prompt rarely fills a 64-row tile (the reference's workload: prompts of
// 1..64 tokens), so a grid of (row, tile-of-the-row) workgroups multiplies mostly padding.  Instead every
// (new row z, token s < L_z) pair gets one flat index; a workgroup takes 64 consecutive pairs, whichever rows they
// belong to.  The host does not know the lengths (they live on the device), so the grid keeps its upper bound and
// every workgroup rebuilds the prefix sums of the new rows' lengths in LDS (<= kMaxCompactRows entries, one
// cooperative scan) and leaves when its tile starts beyond the total.
constexpr int kMaxCompactRows = 2048;

template <int ROWS>
struct FillIndexT {
    int prefix[ROWS + 1];  // prefix[z] = pairs before new row z; prefix[n_new] = total
    int wave_tot[8];
};
using FillIndex = FillIndexT<kMaxCompactRows>;
using NoFillIndex = FillIndexT<1>;  // placeholder of the kernels that do no prefill (zero-length arrays are not allowed)

// The decode projection ("latest" modes) uses the same index with one entry per batch row, worth 1 when the row is
// non-empty: a continuous batch that is 40 % empty slots (a dry page pool, the tail of a run) then multiplies 40 %
// fewer rows.
// all THREADS (a multiple of 64, <= 512) threads of the workgroup call this; returns the total number of pairs
template <int THREADS, bool LATEST, class FI>
__device__ __forceinline__ int build_fill_index(const GemmArgs& g, FI& fi) {
    const int tid = threadIdx.x;
    const int n_entries = LATEST ? g.B : g.n_new;
    const int per = (n_entries + THREADS - 1) / THREADS;  // <= kMaxCompactRows / THREADS
    int sum = 0;
    int local[kMaxCompactRows / THREADS];
#pragma unroll
    for (int j = 0; j < kMaxCompactRows / THREADS; ++j) {
        const int zz = tid * per + j;
        int L = 0;
        if (j < per && zz < n_entries) {
            if (LATEST) L = g.lengths[zz] > 0 ? 1 : 0;
            else L = min(max(g.lengths[g.new_batch_idx[zz]], 0), g.S);
        }
        local[j] = sum;
        sum += L;
    }
    // inclusive scan of `sum` across the wave, then across the waves
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if ((tid & (kWave - 1)) >= off) incl += up;
    }
    if ((tid & (kWave - 1)) == kWave - 1) fi.wave_tot[tid / kWave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / kWave; ++w) {
        if (w < tid / kWave) base += fi.wave_tot[w];
        total += fi.wave_tot[w];
    }
    base += incl - sum;  // exclusive prefix of this thread's first row
#pragma unroll
    for (int j = 0; j < kMaxCompactRows / THREADS; ++j) {
        const int zz = tid * per + j;
        if (j < per && zz < n_entries) fi.prefix[zz] = base + local[j];
    }
    if (tid == 0) fi.prefix[n_entries] = total;
    __syncthreads();
    return total;
}

// flat pair index -> (new row z, token s); i < prefix[n_new]
template <class FI>
__device__ __forceinline__ void fill_index_lookup(const FI& fi, int n_new, int i, int& z, int& s) {
    int lo = 0, hi = n_new;  // largest z with prefix[z] <= i (rows of length 0 share a prefix value: take the last)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (fi.prefix[mid] <= i) lo = mid;
        else hi = mid;
    }
    z = lo;
    s = i - fi.prefix[lo];
}

struct RowDesc {
    const float* a;  // nullptr -> row contributes zeros and is not stored
    float* o;        // BF16 kernels: both point at 16-bit elements and are reinterpreted at the access site
    const float* e;  // embedding prologue (fill modes, g.emb_table set): emb_table row of the token, fp32
    const float* p;  //                                                   wpe row of the position, fp32
};

// bf16 <-> fp32 (bf16 = the upper half of an fp32; products of two bf16 are exact in fp32)
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    // round to nearest even; NaN stays NaN (integer rounding alone would turn some NaNs into Inf/0)
    // (both results, one select: as an early return this is a branch per element in unrolled epilogues)
    const uint32_t u = __float_as_uint(f);
    const uint32_t nan = (u >> 16) | 0x0040u;
    const uint32_t rne = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    return (uint16_t)((u & 0x7fffffffu) > 0x7f800000u ? nan : rne);
}
__device__ __forceinline__ float4 load4_bf16(const void* p) {  // 4 consecutive bf16 (8 bytes)
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                       __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// fp8 (OCP e4m3fn) <-> fp32 for the fp8 page extension.  Decoding is exact (e4m3 is a subset of bf16 and fp32); encoding
// rounds to nearest even and SATURATES at +-448 (the format has no infinity), NaN stays NaN (0x7f).
__device__ __forceinline__ uint32_t f32_to_fp8(float f) {
    if (f != f) return 0x7fu;
    const float c = __builtin_amdgcn_fmed3f(f, -448.f, 448.f);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, c, 0, false) & 0xffu;
}
__device__ __forceinline__ uint32_t f32x4_to_fp8x4(float a, float b, float c, float d) {   // 4 consecutive elements
    return f32_to_fp8(a) | (f32_to_fp8(b) << 8) | (f32_to_fp8(c) << 16) | (f32_to_fp8(d) << 24);
}
// 4 fp8 (one dword) -> 4 bf16 (two dwords): decode to fp32, keep the upper halves (exact)
__device__ __forceinline__ uint2 fp8x4_to_bf16x4(uint32_t w) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    return make_uint2((__float_as_uint(lo.x) >> 16) | (__float_as_uint(lo.y) & 0xffff0000u),
                      (__float_as_uint(hi.x) >> 16) | (__float_as_uint(hi.y) & 0xffff0000u));
}

// BF16: 16-bit page elements; FP8 (with BF16 set: bf16 weights): 8-bit page elements -- same layout rule either way
template <int MODE, bool BF16, bool FP8 = false>
__device__ __forceinline__ RowDesc resolve_row(const GemmArgs& g, int m, int z, int out_id) {
    RowDesc r{nullptr, nullptr, nullptr, nullptr};
    if (MODE == kPlain) {
        if (m < g.M) {
            r.a = g.a_plain + (int64_t)m * g.lda;
            r.o = g.c_plain + (int64_t)m * g.ldc;
        }
        return r;
    }
    int b, s;
    if (MODE == kNaiveLatest || MODE == kPagedLatest) {
        b = m;
        if (b >= g.B) return r;
        const int L = g.lengths[b];
        if (L <= 0) return r;  // empty slot: nothing read, nothing written
        s = L - 1;
    } else {
        // fill: z = index into new_batch_idx, m = token (the compact form resolves the pair before calling)
        b = g.new_batch_idx[z];
        s = m;
        if (s >= g.lengths[b] || s >= g.S) return r;
    }
    if ((MODE == kNaiveFill || MODE == kPagedFill) && g.emb_table != nullptr) {
        r.e = g.emb_table + (int64_t)g.inp[(int64_t)b * g.S + s] * g.K;
        r.p = g.wpe + (int64_t)s * g.K;
    }
    if (MODE == kNaiveLatest || MODE == kNaiveFill) {
        r.a = g.inp_embedding + ((int64_t)b * g.S + s) * g.K;
        if (out_id == 0) {  // K is kept transposed: kt_cache[b, n, s]
            r.o = g.kt_cache + (int64_t)b * g.N * g.S + s;
        } else if (out_id == 1) {
            r.o = g.q_output + (int64_t)b * g.N;
        } else {
            r.o = g.v_cache + ((int64_t)b * g.S + s) * g.N;
        }
    } else {
        float* page = g.page_table[(int64_t)b * (g.S / kPage) + s / kPage];
        if (page == nullptr) return r;  // row longer than its pages (a caller bug): skipped, not dereferenced
        if (FP8) {  // same layout rule, 8-bit elements
            uint8_t* tok = reinterpret_cast<uint8_t*>(page) + page_row_offset(s, g.K, kSegInp);
            r.a = reinterpret_cast<const float*>(tok);
            if (out_id == 1) r.o = g.q_output + (int64_t)b * g.N;  // q stays fp32
            else r.o = reinterpret_cast<float*>(tok + (int64_t)(out_id == 0 ? kSegK : kSegV) * g.K);
        } else if (BF16) {  // same layout rule, 16-bit elements
            uint16_t* tok = reinterpret_cast<uint16_t*>(page) + page_row_offset(s, g.K, kSegInp);
            r.a = reinterpret_cast<const float*>(tok);
            if (out_id == 1) r.o = g.q_output + (int64_t)b * g.N;  // q stays fp32
            else r.o = reinterpret_cast<float*>(tok + (int64_t)(out_id == 0 ? kSegK : kSegV) * g.K);
        } else {
            float* tok = page + page_row_offset(s, g.K, kSegInp);
            r.a = tok;
            if (out_id == 1) r.o = g.q_output + (int64_t)b * g.N;
            else r.o = tok + (int64_t)(out_id == 0 ? kSegK : kSegV) * g.K;
        }
    }
    return r;
}


}  // namespace mli
