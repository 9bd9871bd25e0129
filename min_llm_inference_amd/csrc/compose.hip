// Compositions: the reference's inference_self_attention / paged_attention[_with_cublas] call
// order, issued back to back on one stream.
//   src/kernels/self_attention_inference_optimized.cu:282-301
//   src/kernels/paged_attention.cu:358-377, src/kernels/paged_attention_cublas.cu:260-280
#include "device_common.hpp"

namespace mli {
int launch_latest_naive(const float*, const int*, const float*, const float*, const float*, float*, float*, float*,
                        int, int, int, int, hipStream_t);
int launch_fill_naive(const float*, const int*, const int*, const float*, const float*, float*, float*, int, int, int,
                      int, int, hipStream_t);
int launch_latest_paged(float* const*, const int*, const float*, const float*, const float*, float*, int, int, int,
                        hipStream_t);
int launch_fill_paged(float* const*, const int*, const int*, const float*, const float*, int, int, int, int,
                      hipStream_t);
int launch_qkt_paged(const float*, const float* const*, const int*, float*, int, int, int, hipStream_t);
int launch_qkt_naive(const float*, const float*, const int*, float*, int, int, int, hipStream_t);
int launch_softmax(float*, const int*, int, int, hipStream_t);
int launch_softmax_v_naive(const float*, const float*, const int*, float*, int, int, int, void*, size_t, hipStream_t);
int launch_softmax_v_paged(const float*, const float* const*, const int*, float*, int, int, int, void*, size_t,
                           hipStream_t);
int launch_scores_softmax_v_paged(const float*, const float* const*, const int*, float*, float*, int, int, int, void*,
                                  size_t, hipStream_t);
int launch_scores_softmax_v_naive(const float*, const float*, const float*, const int*, float*, float*, int, int, int,
                                  void*, size_t, hipStream_t);
int launch_scores_softmax_v_paged_bf16(const float*, const uint16_t* const*, const int*, float*, float*, int, int, int,
                                       void*, size_t, hipStream_t);
// single-pass fused scan (attention_fused.hip): 1 = ran, 0 = not applicable (caller falls back), else error + (rc > 0)
int launch_fused_decode_f32(const float*, const float* const*, const int*, float*, float*, int, int, int, void*, size_t,
                            hipStream_t);
int launch_fused_decode_bf16(const float*, const uint16_t* const*, const int*, float*, float*, int, int, int, void*,
                             size_t, hipStream_t);
int launch_latest_paged_bf16(uint16_t* const*, const int*, const uint16_t*, const uint16_t*, const uint16_t*, float*, int,
                             int, int, hipStream_t);
int launch_fill_paged_bf16(uint16_t* const*, const int*, const int*, const uint16_t*, const uint16_t*, int, int, int,
                           int, hipStream_t);
int launch_qkt_paged_bf16(const float*, const uint16_t* const*, const int*, float*, int, int, int, hipStream_t);
int launch_softmax_v_paged_bf16(const float*, const uint16_t* const*, const int*, float*, int, int, int, void*, size_t,
                                hipStream_t);
}  // namespace mli

extern "C" {

int mli_abi_version(void) { return 2; }

int mli_paged_attention_bf16(mli_bf16* const* page_table, const int* lengths, const mli_bf16* wk, const mli_bf16* wq,
                             const mli_bf16* wv, const int* new_batch_idx, float* q_output, float* qkt_output,
                             float* attention_result, int n_batch, int n_sequence, int emb_dim, int n_new_items,
                             void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_fill_paged_bf16(page_table, new_batch_idx, lengths, wk, wv, n_batch, n_sequence, emb_dim,
                                         n_new_items, st);
    if (rc) return rc;
    rc = mli::launch_latest_paged_bf16(page_table, lengths, wk, wq, wv, q_output, n_batch, n_sequence, emb_dim, st);
    if (rc) return rc;
    const int fused = mli::launch_fused_decode_bf16(q_output, page_table, lengths, qkt_output, attention_result, n_batch,
                                                    n_sequence, emb_dim, workspace, workspace_bytes, st);
    if (fused == 1) return 0;
    if (fused != 0) return fused < 0 ? fused : fused - 1;
    return mli::launch_scores_softmax_v_paged_bf16(q_output, page_table, lengths, qkt_output, attention_result,
                                                   n_batch, n_sequence, emb_dim, workspace, workspace_bytes, st);
}

int mli_inference_self_attention(const float* inp_embedding, const int* lengths, const float* wk, const float* wq,
                                 const float* wv, const int* new_batch_idx, float* kt_cache, float* v_cache,
                                 float* q_output, float* qkt_output, float* attention_result, int n_batch,
                                 int n_sequence, int input_dim, int output_dim, int n_new_items, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_fill_naive(inp_embedding, new_batch_idx, lengths, wk, wv, kt_cache, v_cache, n_batch,
                                    n_sequence, input_dim, output_dim, n_new_items, st);
    if (rc) return rc;
    rc = mli::launch_latest_naive(inp_embedding, lengths, wk, wq, wv, kt_cache, v_cache, q_output, n_batch,
                                  n_sequence, input_dim, output_dim, st);
    if (rc) return rc;
    return mli::launch_scores_softmax_v_naive(q_output, kt_cache, v_cache, lengths, qkt_output, attention_result,
                                              n_batch, n_sequence, output_dim, workspace, workspace_bytes, st);
}

int mli_paged_attention(float* const* page_table, const int* lengths, const float* wk, const float* wq,
                        const float* wv, const int* new_batch_idx, float* q_output, float* qkt_output,
                        float* attention_result, int n_batch, int n_sequence, int emb_dim, int n_new_items,
                        void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_fill_paged(page_table, new_batch_idx, lengths, wk, wv, n_batch, n_sequence, emb_dim,
                                    n_new_items, st);
    if (rc) return rc;
    rc = mli::launch_latest_paged(page_table, lengths, wk, wq, wv, q_output, n_batch, n_sequence, emb_dim, st);
    if (rc) return rc;
    const int fused = mli::launch_fused_decode_f32(q_output, page_table, lengths, qkt_output, attention_result, n_batch,
                                                   n_sequence, emb_dim, workspace, workspace_bytes, st);
    if (fused == 1) return 0;
    if (fused != 0) return fused < 0 ? fused : fused - 1;
    return mli::launch_scores_softmax_v_paged(q_output, page_table, lengths, qkt_output, attention_result, n_batch,
                                              n_sequence, emb_dim, workspace, workspace_bytes, st);
}

}  // extern "C"
