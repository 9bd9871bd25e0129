// Token embedding for newly inserted rows and the greedy decoder head, plus the
// contiguous->paged cloning helper the parity tests use.  Reference behaviour:
//   src/kernels/encoder.cu:56-147  (inference_optimized_encoder, paged_attention_encoder)
//   src/kernels/decoder.cu:25-255  (decoder_kernel, paged_attention_decoder_kernel_with_multi_decoder)
//   src/kernels/utils.cu:106-160   (clone_inp_embedding_k_v_cache)
#include <cfloat>

#include "gemm_common.hpp"

namespace mli {

int launch_gemm_nt(const float* A, const float* Bt, float* C, int M, int N, int K, hipStream_t st);
int launch_gemm_nt_argmax(const float* A, const float* Bt, RowBest* row_best, int M, int N, int K, int* n_tiles,
                          hipStream_t st);
int gemm_nt_argmax_max_tiles(int N);

constexpr int kEdThreads = 256;

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    auto cvt = [](float f) -> uint32_t {  // round to nearest even, NaN preserved
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x0040u;
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    return cvt(lo) | (cvt(hi) << 16);
}

// emb + wpe of 4 consecutive columns -> fp32 row, bf16 row or fp8 (OCP e4m3) row; ELEM = MLI_ELEM_* (false / true = f32 / bf16)
template <int ELEM>
__device__ __forceinline__ void store_sum4(float* dst_row, int i4, const float4& a, const float4& c) {
    const float4 r = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    if (ELEM == MLI_ELEM_FP8) reinterpret_cast<uint32_t*>(dst_row)[i4] = f32x4_to_fp8x4(r.x, r.y, r.z, r.w);
    else if (ELEM == MLI_ELEM_BF16) reinterpret_cast<uint2*>(dst_row)[i4] = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
    else reinterpret_cast<float4*>(dst_row)[i4] = r;
}

// the row of position s in segment `seg` of a page whose elements are ELEM
template <int ELEM>
__device__ __forceinline__ float* page_row_ptr(float* page, int s, int D, int seg) {
    const int64_t off = page_row_offset(s, D, seg);
    if (ELEM == MLI_ELEM_FP8) return reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(page) + off);
    if (ELEM == MLI_ELEM_BF16) return reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(page) + off);
    return page + off;
}

// One workgroup per (16-token group, new row); one wave per token, lanes along the embedding.
template <bool PAGED, int BF16 = 0>
__global__ __launch_bounds__(kEdThreads) void encoder_new_rows_kernel(
    const float* __restrict__ emb_table, const float* __restrict__ wpe, const int* __restrict__ inp,
    float* __restrict__ inp_embedding, float* const* __restrict__ page_table, const int* __restrict__ lengths,
    const int* __restrict__ new_item_indices, int S, int D) {
    __shared__ float* page_sh;
    const int b = new_item_indices[blockIdx.y];
    const int L = lengths[b];
    const int s_base = blockIdx.x * kPage;
    if (s_base >= L) return;
    if (PAGED) {
        if (threadIdx.x == 0) page_sh = page_table[(int64_t)b * (S / kPage) + blockIdx.x];
        __syncthreads();
        if (page_sh == nullptr) return;  // a caller bug (row longer than its pages): skip rather than fault the GPU
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int D4 = D >> 2;
    for (int t = wave; t < kPage; t += kEdThreads / kWave) {
        const int s = s_base + t;
        if (s >= L) break;
        const int tok = inp[(int64_t)b * S + s];
        const float4* e = reinterpret_cast<const float4*>(emb_table + (int64_t)tok * D);
        const float4* p = reinterpret_cast<const float4*>(wpe + (int64_t)s * D);
        float* dst = !PAGED ? inp_embedding + ((int64_t)b * S + s) * D : page_row_ptr<BF16>(page_sh, s, D, kSegInp);
        for (int i = lane; i < D4; i += kWave) store_sum4<BF16>(dst, i, e[i], p[i]);
    }
}

// One workgroup per batch row.  argmax keeps the LOWEST index among equal maxima (the reference's
// host decoder, tests/test_utils.cpp:607-614; its device kernel breaks ties by thread order).
template <bool PAGED, int BF16 = 0>
__global__ __launch_bounds__(kEdThreads) void decoder_argmax_kernel(
    const float* __restrict__ emb_score, int* __restrict__ decoder_result, int* __restrict__ lengths,
    float* __restrict__ inp_embedding, float* const* __restrict__ page_table, const float* __restrict__ wpe_table,
    const float* __restrict__ emb_table, int n_vocab, int S, int D, int n_decoder_results, int i_decoder) {
    __shared__ float best_v[kEdThreads / kWave];
    __shared__ int best_i[kEdThreads / kWave];
    __shared__ int tok_sh;
    __shared__ float* page_sh;
    const int b = blockIdx.x;
    const int L = lengths[b];
    if (L == 0) {  // empty slot
        if (threadIdx.x == 0) decoder_result[(int64_t)b * n_decoder_results + i_decoder] = MLI_EMPTY_ROW_TOKEN_ID;
        return;
    }
    const float* sc = emb_score + (int64_t)b * n_vocab;
    float mv = -FLT_MAX;
    int mi = -1;
    for (int i = threadIdx.x; i < n_vocab; i += kEdThreads) {
        const float v = sc[i];
        if (v > mv) { mv = v; mi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(mv, off, kWave);
        const int oi = __shfl_xor(mi, off, kWave);
        if (ov > mv || (ov == mv && (unsigned)oi < (unsigned)mi)) { mv = ov; mi = oi; }
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (lane == 0) { best_v[wave] = mv; best_i[wave] = mi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = best_v[0];
        int bi = best_i[0];
        for (int w = 1; w < kEdThreads / kWave; ++w)
            if (best_v[w] > bv || (best_v[w] == bv && (unsigned)best_i[w] < (unsigned)bi)) { bv = best_v[w]; bi = best_i[w]; }
        tok_sh = bi;
        decoder_result[(int64_t)b * n_decoder_results + i_decoder] = bi;
        const bool done = (L + 1 >= S) || bi == MLI_EOF_TOKEN_ID;
        lengths[b] = done ? 0 : L + 1;
        if (PAGED && !done) page_sh = page_table[(int64_t)b * (S / kPage) + L / kPage];
    }
    __syncthreads();
    const int tok = tok_sh;
    if (L + 1 >= S || tok == MLI_EOF_TOKEN_ID || tok < 0) return;  // finished rows get no next embedding
    if (PAGED && page_sh == nullptr) return;  // no page for the next position (a caller bug): skip rather than fault
    const float4* e = reinterpret_cast<const float4*>(emb_table + (int64_t)tok * D);
    const float4* p = reinterpret_cast<const float4*>(wpe_table + (int64_t)L * D);
    float* dst = !PAGED ? inp_embedding + ((int64_t)b * S + L) * D : page_row_ptr<BF16>(page_sh, L, D, kSegInp);
    for (int i = threadIdx.x; i < (D >> 2); i += kEdThreads) store_sum4<BF16>(dst, i, e[i], p[i]);
}

// The decoder head behind the logits GEMM's argmax epilogue (proj_gemm.hip: launch_gemm_nt_argmax): one WAVE per batch
// row picks the row's best (value, index) pair among the column tiles' pairs -- larger value, then lower index, the
// order decoder_argmax_kernel applies to the scores themselves, so the token is the same -- then updates the length
// and writes the next input embedding exactly as decoder_argmax_kernel does (reference decoder.cu:128-205).
constexpr int kFinalizeRows = kEdThreads / kWave;  // batch rows per workgroup

template <bool PAGED, int BF16 = 0>
__global__ __launch_bounds__(kEdThreads) void decoder_finalize_kernel(
    const RowBest* __restrict__ row_best, int n_tiles, int* __restrict__ decoder_result, int* __restrict__ lengths,
    float* __restrict__ inp_embedding, float* const* __restrict__ page_table, const float* __restrict__ wpe_table,
    const float* __restrict__ emb_table, int B, int S, int D, int n_decoder_results, int i_decoder) {
    const int lane = threadIdx.x & (kWave - 1);
    const int b = blockIdx.x * kFinalizeRows + (threadIdx.x >> 6);
    if (b >= B) return;
    const int L = lengths[b];
    if (L == 0) {  // empty slot
        if (lane == 0) decoder_result[(int64_t)b * n_decoder_results + i_decoder] = MLI_EMPTY_ROW_TOKEN_ID;
        return;
    }
    // the page of the next position depends on the length only: requested together with the row's pairs, not behind
    // the token (one dependent memory round trip less in a kernel that is nothing but such round trips)
    float* page = nullptr;
    if (PAGED && L + 1 < S) page = page_table[(int64_t)b * (S / kPage) + L / kPage];  // same address in every lane: one request
    float mv = -FLT_MAX;
    int mi = -1;
    for (int t = lane; t < n_tiles; t += kWave) {
        const RowBest rb = row_best[(int64_t)b * n_tiles + t];
        argmax_take(mv, mi, rb.value, rb.index);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(mv, off, kWave);
        const int oi = __shfl_xor(mi, off, kWave);
        argmax_take(mv, mi, ov, oi);
    }
    const int tok = mi;  // every lane holds the same pair now
    const bool done = (L + 1 >= S) || tok == MLI_EOF_TOKEN_ID;
    if (lane == 0) {
        decoder_result[(int64_t)b * n_decoder_results + i_decoder] = tok;
        lengths[b] = done ? 0 : L + 1;
    }
    if (done || tok < 0) return;               // finished rows get no next embedding
    if (PAGED && page == nullptr) return;      // no page for the next position (a caller bug): skip rather than fault
    const float4* e = reinterpret_cast<const float4*>(emb_table + (int64_t)tok * D);
    const float4* p = reinterpret_cast<const float4*>(wpe_table + (int64_t)L * D);
    float* dst = !PAGED ? inp_embedding + ((int64_t)b * S + L) * D : page_row_ptr<BF16>(page, L, D, kSegInp);
    for (int i = lane; i < (D >> 2); i += kWave) store_sum4<BF16>(dst, i, e[i], p[i]);
}

// fp32 -> fp8 (OCP e4m3, round to nearest even, saturating) with the page kernels' own conversion: test / bench support
__global__ __launch_bounds__(kEdThreads) void f32_to_fp8_kernel(const float4* __restrict__ src, uint32_t* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * kEdThreads + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * kEdThreads;
    for (; i < n4; i += stride) {
        const float4 v = src[i];
        dst[i] = f32x4_to_fp8x4(v.x, v.y, v.z, v.w);
    }
}

// grid = (S/16, B).  Positions 0..min(L, S-1) inclusive are cloned (the decoder writes the next
// position's embedding, so one more than the length is materialised; reference utils.cu:125-127).
__global__ __launch_bounds__(kEdThreads) void clone_to_pages_kernel(
    float* const* __restrict__ page_table, const float* __restrict__ inp_embedding,
    const float* __restrict__ kt_cache, const float* __restrict__ v_cache, const int* __restrict__ lengths,
    int S, int D) {
    const int b = blockIdx.y;
    const int L = lengths[b];
    if (L == 0) return;  // never allocated
    const int last = min(L, S - 1);
    const int s_base = blockIdx.x * kPage;
    if (s_base > last) return;
    float* page = page_table[(int64_t)b * (S / kPage) + blockIdx.x];
    if (page == nullptr) return;
    for (int idx = threadIdx.x; idx < kPage * D; idx += kEdThreads) {
        const int t = idx / D, d = idx % D;
        const int s = s_base + t;
        if (s > last) continue;
        float* row = page + (int64_t)t * 3 * D;
        row[d] = inp_embedding[((int64_t)b * S + s) * D + d];
        row[D + d] = kt_cache[((int64_t)b * D + d) * S + s];
        row[2 * D + d] = v_cache[((int64_t)b * S + s) * D + d];
    }
}

}  // namespace mli

extern "C" {

int mli_inference_optimized_encoder(const float* emb_table, const float* wpe, const int* inp, float* inp_embedding,
                                    const int* lengths, const int* new_item_indices, int n_batch, int n_sequence,
                                    int emb_dim, int n_new_items, void* stream) {
    if (n_new_items == 0) return 0;  // reference encoder.cu:84-86
    if (n_new_items < 0 || emb_dim % 4 != 0 || n_batch <= 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL((mli::encoder_new_rows_kernel<false>), dim3(mli::ceil_div_i(n_sequence, mli::kPage), n_new_items),
                       dim3(mli::kEdThreads), 0, mli::as_stream(stream), emb_table, wpe, inp, inp_embedding,
                       (float* const*)nullptr, lengths, new_item_indices, n_sequence, emb_dim);
    return mli::launch_status();
}

int mli_paged_attention_encoder(const float* emb_table, const float* wpe, const int* inp, float* const* page_table,
                                const int* lengths, const int* new_item_indices, int n_batch, int n_sequence,
                                int emb_dim, int n_new_items, void* stream) {
    if (n_new_items == 0) return 0;  // reference encoder.cu:138-140
    if (n_new_items < 0 || emb_dim % 4 != 0 || n_sequence % mli::kPage != 0 || n_batch <= 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL((mli::encoder_new_rows_kernel<true>), dim3(n_sequence / mli::kPage, n_new_items),
                       dim3(mli::kEdThreads), 0, mli::as_stream(stream), emb_table, wpe, inp, (float*)nullptr,
                       page_table, lengths, new_item_indices, n_sequence, emb_dim);
    return mli::launch_status();
}

int mli_paged_attention_encoder_bf16(const float* emb_table, const float* wpe, const int* inp,
                                     mli_bf16* const* page_table, const int* lengths, const int* new_item_indices,
                                     int n_batch, int n_sequence, int emb_dim, int n_new_items, void* stream) {
    if (n_new_items == 0) return 0;
    if (n_new_items < 0 || emb_dim % 8 != 0 || n_sequence % mli::kPage != 0 || n_batch <= 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL((mli::encoder_new_rows_kernel<true, MLI_ELEM_BF16>), dim3(n_sequence / mli::kPage, n_new_items),
                       dim3(mli::kEdThreads), 0, mli::as_stream(stream), emb_table, wpe, inp, (float*)nullptr,
                       reinterpret_cast<float* const*>(page_table), lengths, new_item_indices, n_sequence, emb_dim);
    return mli::launch_status();
}

int mli_paged_decoder_multi_rounds_bf16(const float* batch_result, const float* emb_table, float* emb_score,
                                        const float* wpe_table, mli_bf16* const* page_table, int* lengths,
                                        int* decoder_result, int n_batch, int n_vocab, int n_sequence, int emb_dim,
                                        int n_decoder_results, int i_decoder, void* stream) {
    if (emb_dim % 8 != 0 || n_sequence % mli::kPage != 0 || n_batch <= 0 || n_vocab <= 0 || n_decoder_results <= 0 ||
        i_decoder < 0 || i_decoder >= n_decoder_results)
        return MLI_ERR_BAD_ARG;
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_gemm_nt(batch_result, emb_table, emb_score, n_batch, n_vocab, emb_dim, st);
    if (rc) return rc;
    hipLaunchKernelGGL((mli::decoder_argmax_kernel<true, MLI_ELEM_BF16>), dim3(n_batch), dim3(mli::kEdThreads), 0, st,
                       emb_score, decoder_result, lengths, (float*)nullptr,
                       reinterpret_cast<float* const*>(page_table), wpe_table, emb_table, n_vocab, n_sequence,
                       emb_dim, n_decoder_results, i_decoder);
    return mli::launch_status();
}

int mli_decoder(const float* batch_result, const float* emb_table, float* emb_score, const float* wpe_table,
                float* inp_embedding, int* lengths, int* decoder_result, int n_batch, int n_vocab, int n_sequence,
                int emb_dim, void* stream) {
    if (emb_dim % 4 != 0 || n_batch <= 0 || n_vocab <= 0) return MLI_ERR_BAD_ARG;
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_gemm_nt(batch_result, emb_table, emb_score, n_batch, n_vocab, emb_dim, st);
    if (rc) return rc;
    hipLaunchKernelGGL((mli::decoder_argmax_kernel<false>), dim3(n_batch), dim3(mli::kEdThreads), 0, st, emb_score,
                       decoder_result, lengths, inp_embedding, (float* const*)nullptr, wpe_table, emb_table, n_vocab,
                       n_sequence, emb_dim, 1, 0);
    return mli::launch_status();
}

int mli_paged_decoder_multi_rounds(const float* batch_result, const float* emb_table, float* emb_score,
                                   const float* wpe_table, float* const* page_table, int* lengths,
                                   int* decoder_result, int n_batch, int n_vocab, int n_sequence, int emb_dim,
                                   int n_decoder_results, int i_decoder, void* stream) {
    if (emb_dim % 4 != 0 || n_sequence % mli::kPage != 0 || n_batch <= 0 || n_vocab <= 0 || n_decoder_results <= 0 ||
        i_decoder < 0 || i_decoder >= n_decoder_results)
        return MLI_ERR_BAD_ARG;
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_gemm_nt(batch_result, emb_table, emb_score, n_batch, n_vocab, emb_dim, st);
    if (rc) return rc;
    hipLaunchKernelGGL((mli::decoder_argmax_kernel<true>), dim3(n_batch), dim3(mli::kEdThreads), 0, st, emb_score,
                       decoder_result, lengths, (float*)nullptr, page_table, wpe_table, emb_table, n_vocab,
                       n_sequence, emb_dim, n_decoder_results, i_decoder);
    return mli::launch_status();
}

size_t mli_decoder_scratch_bytes(int n_batch, int n_vocab) {
    if (n_batch <= 0 || n_vocab <= 0) return 0;
    return (size_t)n_batch * mli::gemm_nt_argmax_max_tiles(n_vocab) * sizeof(mli::RowBest);
}

// layout: 0 = contiguous (inp_embedding), 1 = paged fp32, 2 = paged bf16, 3 = paged fp8
static int decoder_fused(int layout, const float* batch_result, const float* emb_table, const float* wpe_table,
                         float* inp_embedding, float* const* page_table, int* lengths, int* decoder_result, int n_batch,
                         int n_vocab, int n_sequence, int emb_dim, int n_decoder_results, int i_decoder, void* scratch,
                         size_t scratch_bytes, void* stream) {
    if (emb_dim % (layout == 3 ? 16 : layout == 2 ? 8 : 4) != 0 || n_batch <= 0 || n_vocab <= 0 || n_decoder_results <= 0 || i_decoder < 0 ||
        i_decoder >= n_decoder_results || (layout != 0 && n_sequence % mli::kPage != 0))
        return MLI_ERR_BAD_ARG;
    if (scratch == nullptr || scratch_bytes < mli_decoder_scratch_bytes(n_batch, n_vocab)) return MLI_ERR_WORKSPACE;
    hipStream_t st = mli::as_stream(stream);
    mli::RowBest* best = reinterpret_cast<mli::RowBest*>(scratch);
    int n_tiles = 0;
    int rc = mli::launch_gemm_nt_argmax(batch_result, emb_table, best, n_batch, n_vocab, emb_dim, &n_tiles, st);
    if (rc) return rc;
    const dim3 grid(mli::ceil_div_i(n_batch, mli::kFinalizeRows)), block(mli::kEdThreads);
    if (layout == 0)
        hipLaunchKernelGGL((mli::decoder_finalize_kernel<false>), grid, block, 0, st, best, n_tiles, decoder_result, lengths,
                           inp_embedding, (float* const*)nullptr, wpe_table, emb_table, n_batch, n_sequence, emb_dim,
                           n_decoder_results, i_decoder);
    else if (layout == 1)
        hipLaunchKernelGGL((mli::decoder_finalize_kernel<true>), grid, block, 0, st, best, n_tiles, decoder_result, lengths,
                           (float*)nullptr, page_table, wpe_table, emb_table, n_batch, n_sequence, emb_dim,
                           n_decoder_results, i_decoder);
    else if (layout == 2)
        hipLaunchKernelGGL((mli::decoder_finalize_kernel<true, MLI_ELEM_BF16>), grid, block, 0, st, best, n_tiles, decoder_result,
                           lengths, (float*)nullptr, page_table, wpe_table, emb_table, n_batch, n_sequence, emb_dim,
                           n_decoder_results, i_decoder);
    else
        hipLaunchKernelGGL((mli::decoder_finalize_kernel<true, MLI_ELEM_FP8>), grid, block, 0, st, best, n_tiles, decoder_result,
                           lengths, (float*)nullptr, page_table, wpe_table, emb_table, n_batch, n_sequence, emb_dim,
                           n_decoder_results, i_decoder);
    return mli::launch_status();
}

int mli_decoder_fused(const float* batch_result, const float* emb_table, const float* wpe_table, float* inp_embedding,
                      int* lengths, int* decoder_result, int n_batch, int n_vocab, int n_sequence, int emb_dim,
                      void* scratch, size_t scratch_bytes, void* stream) {
    return decoder_fused(0, batch_result, emb_table, wpe_table, inp_embedding, nullptr, lengths, decoder_result, n_batch,
                         n_vocab, n_sequence, emb_dim, 1, 0, scratch, scratch_bytes, stream);
}

int mli_paged_decoder_fused(const float* batch_result, const float* emb_table, const float* wpe_table,
                            void* const* page_table, int* lengths, int* decoder_result, int n_batch, int n_vocab,
                            int n_sequence, int emb_dim, int n_decoder_results, int i_decoder, int elem_bf16, void* scratch,
                            size_t scratch_bytes, void* stream) {
    if (elem_bf16 < MLI_ELEM_F32 || elem_bf16 > MLI_ELEM_FP8) return MLI_ERR_BAD_ARG;
    return decoder_fused(1 + elem_bf16, batch_result, emb_table, wpe_table, nullptr,
                         reinterpret_cast<float* const*>(page_table), lengths, decoder_result, n_batch, n_vocab,
                         n_sequence, emb_dim, n_decoder_results, i_decoder, scratch, scratch_bytes, stream);
}

int mli_f32_to_fp8(const float* src, uint8_t* dst, size_t n, void* stream) {
    if (n % 4 != 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL(mli::f32_to_fp8_kernel, dim3(1024), dim3(mli::kEdThreads), 0, mli::as_stream(stream),
                       reinterpret_cast<const float4*>(src), reinterpret_cast<uint32_t*>(dst), n / 4);
    return mli::launch_status();
}

int mli_clone_inp_embedding_k_v_cache(float* const* page_table, const float* inp_embedding, const float* kt_cache,
                                      const float* v_cache, const int* lengths, int n_batch, int n_sequence,
                                      int emb_dim, void* stream) {
    if (n_sequence % mli::kPage != 0 || n_batch <= 0) return MLI_ERR_BAD_ARG;
    hipLaunchKernelGGL(mli::clone_to_pages_kernel, dim3(n_sequence / mli::kPage, n_batch), dim3(mli::kEdThreads), 0,
                       mli::as_stream(stream), page_table, inp_embedding, kt_cache, v_cache, lengths, n_sequence,
                       emb_dim);
    return mli::launch_status();
}

}  // extern "C"
