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
int launch_fill_paged_embed(const float*, const float*, const int*, float* const*, const int*, const int*, const float*,
                            const float*, int, int, int, int, hipStream_t);
int launch_fill_paged_bf16_embed(const float*, const float*, const int*, uint16_t* const*, const int*, const int*,
                                 const uint16_t*, const uint16_t*, int, int, int, int, hipStream_t);
bool prefill_fuses(int emb_dim);   // proj_gemm.hip: mli_tune "prefill_fused"
// single-launch scan over the contiguous caches (attention_fused_naive.hip): 1 = ran, 0 = shape not covered, else error + (rc > 0)
int launch_fused_decode_naive(const float*, const float*, const float*, const int*, float*, int, int, int, void*, size_t,
                              hipStream_t);
int launch_qkt_paged_bf16(const float*, const uint16_t* const*, const int*, float*, int, int, int, hipStream_t);
// fp8 (OCP e4m3) pages with bf16 weights: MLI_ELEM_FP8 of the lean entry points
int launch_latest_paged_fp8(uint8_t* const*, const int*, const uint16_t*, const uint16_t*, const uint16_t*, float*, int, int,
                            int, hipStream_t);
int launch_fill_paged_fp8_embed(const float*, const float*, const int*, uint8_t* const*, const int*, const int*,
                                const uint16_t*, const uint16_t*, int, int, int, int, hipStream_t);
int launch_fused_decode_fp8(const float*, const uint8_t* const*, const int*, float*, int, int, int, void*, size_t, hipStream_t);
int launch_softmax_v_paged_bf16(const float*, const uint16_t* const*, const int*, float*, int, int, int, void*, size_t,
                                hipStream_t);
}  // namespace mli

extern "C" {

int mli_abi_version(void) { return 4; }

int mli_elem_supported(int elem) { return elem == MLI_ELEM_F32 || elem == MLI_ELEM_BF16 || elem == MLI_ELEM_FP8; }

int mli_paged_attention_lean(void* const* page_table, const int* lengths, const void* wk, const void* wq, const void* wv,
                             const int* new_batch_idx, float* q_output, float* attention_result, int n_batch,
                             int n_sequence, int emb_dim, int n_new_items, int elem_bf16, void* workspace,
                             size_t workspace_bytes, void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
    hipStream_t st = mli::as_stream(stream);
    int rc, fused;
    if (elem_bf16 < MLI_ELEM_F32 || elem_bf16 > MLI_ELEM_FP8) return MLI_ERR_BAD_ARG;
    if (elem_bf16 == MLI_ELEM_FP8) {
        uint8_t* const* pt = reinterpret_cast<uint8_t* const*>(page_table);
        const mli_bf16 *k = static_cast<const mli_bf16*>(wk), *q = static_cast<const mli_bf16*>(wq), *v = static_cast<const mli_bf16*>(wv);
        rc = mli::launch_fill_paged_fp8_embed(nullptr, nullptr, nullptr, pt, new_batch_idx, lengths, k, v, n_batch, n_sequence,
                                              emb_dim, n_new_items, st);
        if (!rc) rc = mli::launch_latest_paged_fp8(pt, lengths, k, q, v, q_output, n_batch, n_sequence, emb_dim, st);
        if (rc) return rc;
        fused = mli::launch_fused_decode_fp8(q_output, pt, lengths, attention_result, n_batch, n_sequence, emb_dim, workspace,
                                             workspace_bytes, st);
    } else if (elem_bf16) {
        mli_bf16* const* pt = reinterpret_cast<mli_bf16* const*>(page_table);
        const mli_bf16 *k = static_cast<const mli_bf16*>(wk), *q = static_cast<const mli_bf16*>(wq), *v = static_cast<const mli_bf16*>(wv);
        rc = mli::launch_fill_paged_bf16(pt, new_batch_idx, lengths, k, v, n_batch, n_sequence, emb_dim, n_new_items, st);
        if (!rc) rc = mli::launch_latest_paged_bf16(pt, lengths, k, q, v, q_output, n_batch, n_sequence, emb_dim, st);
        if (rc) return rc;
        fused = mli::launch_fused_decode_bf16(q_output, pt, lengths, nullptr, attention_result, n_batch, n_sequence,
                                              emb_dim, workspace, workspace_bytes, st);
    } else {
        float* const* pt = reinterpret_cast<float* const*>(page_table);
        const float *k = static_cast<const float*>(wk), *q = static_cast<const float*>(wq), *v = static_cast<const float*>(wv);
        rc = mli::launch_fill_paged(pt, new_batch_idx, lengths, k, v, n_batch, n_sequence, emb_dim, n_new_items, st);
        if (!rc) rc = mli::launch_latest_paged(pt, lengths, k, q, v, q_output, n_batch, n_sequence, emb_dim, st);
        if (rc) return rc;
        fused = mli::launch_fused_decode_f32(q_output, pt, lengths, nullptr, attention_result, n_batch, n_sequence,
                                             emb_dim, workspace, workspace_bytes, st);
    }
    if (fused == 1) return 0;
    // rows too wide for the single-pass kernel (or no workspace): the caller takes the materialising composition
    if (fused == 0) return MLI_ERR_BAD_ARG;
    return fused < 0 ? fused : fused - 1;
}

int mli_get_latest_k_q_v_paged_lean(void* const* page_table, const int* lengths, const void* wk, const void* wq,
                                    const void* wv, float* q_output, int n_batch, int n_sequence, int emb_dim, int elem,
                                    void* stream) {
    hipStream_t st = mli::as_stream(stream);
    if (elem == MLI_ELEM_FP8)
        return mli::launch_latest_paged_fp8(reinterpret_cast<uint8_t* const*>(page_table), lengths, static_cast<const mli_bf16*>(wk),
                                            static_cast<const mli_bf16*>(wq), static_cast<const mli_bf16*>(wv), q_output, n_batch,
                                            n_sequence, emb_dim, st);
    if (elem == MLI_ELEM_BF16)
        return mli::launch_latest_paged_bf16(reinterpret_cast<mli_bf16* const*>(page_table), lengths, static_cast<const mli_bf16*>(wk),
                                             static_cast<const mli_bf16*>(wq), static_cast<const mli_bf16*>(wv), q_output, n_batch,
                                             n_sequence, emb_dim, st);
    if (elem == MLI_ELEM_F32)
        return mli::launch_latest_paged(reinterpret_cast<float* const*>(page_table), lengths, static_cast<const float*>(wk),
                                        static_cast<const float*>(wq), static_cast<const float*>(wv), q_output, n_batch,
                                        n_sequence, emb_dim, st);
    return MLI_ERR_BAD_ARG;
}

int mli_self_attention_lean(const float* inp_embedding, const int* lengths, const float* wk, const float* wq,
                            const float* wv, const int* new_batch_idx, float* kt_cache, float* v_cache, float* q_output,
                            float* attention_result, int n_batch, int n_sequence, int input_dim, int output_dim,
                            int n_new_items, void* workspace, size_t workspace_bytes, void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
    hipStream_t st = mli::as_stream(stream);
    int rc = mli::launch_fill_naive(inp_embedding, new_batch_idx, lengths, wk, wv, kt_cache, v_cache, n_batch,
                                    n_sequence, input_dim, output_dim, n_new_items, st);
    if (rc) return rc;
    rc = mli::launch_latest_naive(inp_embedding, lengths, wk, wq, wv, kt_cache, v_cache, q_output, n_batch,
                                  n_sequence, input_dim, output_dim, st);
    if (rc) return rc;
    const int fused = mli::launch_fused_decode_naive(q_output, kt_cache, v_cache, lengths, attention_result, n_batch,
                                                     n_sequence, output_dim, workspace, workspace_bytes, st);
    if (fused == 1) return 0;
    // shapes the single-launch scan does not cover (dims not multiples of 4, no workspace): the caller takes the
    // materialising composition -- the projection it has just run is idempotent
    if (fused == 0) return MLI_ERR_BAD_ARG;
    return fused < 0 ? fused : fused - 1;
}

int mli_paged_decode_step(void* const* page_table, int* lengths, const void* wk, const void* wq, const void* wv,
                          const float* emb_table, const float* wpe_table, float* q_output, float* attention_result,
                          int* decoder_result, int n_batch, int n_sequence, int emb_dim, int n_vocab,
                          int n_decoder_results, int i_decoder, int elem_bf16, void* workspace, size_t workspace_bytes,
                          void* decoder_scratch, size_t decoder_scratch_bytes, void* stream) {
    const int rc = mli_paged_attention_lean(page_table, lengths, wk, wq, wv, /*new_batch_idx=*/nullptr, q_output,
                                            attention_result, n_batch, n_sequence, emb_dim, /*n_new_items=*/0, elem_bf16,
                                            workspace, workspace_bytes, stream);
    if (rc) return rc;
    return mli_paged_decoder_fused(attention_result, emb_table, wpe_table, page_table, lengths, decoder_result, n_batch,
                                   n_vocab, n_sequence, emb_dim, n_decoder_results, i_decoder, elem_bf16, decoder_scratch,
                                   decoder_scratch_bytes, stream);
}

int mli_decode_step(float* inp_embedding, int* lengths, const float* wk, const float* wq, const float* wv,
                    const float* emb_table, const float* wpe_table, float* kt_cache, float* v_cache, float* q_output,
                    float* qkt_output, float* attention_result, int* decoder_result, int n_batch, int n_sequence,
                    int emb_dim, int n_vocab, void* workspace, size_t workspace_bytes, void* decoder_scratch,
                    size_t decoder_scratch_bytes, void* stream) {
    int rc = mli_self_attention_lean(inp_embedding, lengths, wk, wq, wv, /*new_batch_idx=*/nullptr, kt_cache, v_cache,
                                     q_output, attention_result, n_batch, n_sequence, emb_dim, emb_dim, /*n_new_items=*/0,
                                     workspace, workspace_bytes, stream);
    if (rc == MLI_ERR_BAD_ARG)  // a shape the single-launch scan does not cover: scores and probabilities through qkt_output
        rc = mli_inference_self_attention(inp_embedding, lengths, wk, wq, wv, /*new_batch_idx=*/nullptr, kt_cache, v_cache,
                                          q_output, qkt_output, attention_result, n_batch, n_sequence, emb_dim, emb_dim,
                                          /*n_new_items=*/0, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return mli_decoder_fused(attention_result, emb_table, wpe_table, inp_embedding, lengths, decoder_result, n_batch,
                             n_vocab, n_sequence, emb_dim, decoder_scratch, decoder_scratch_bytes, stream);
}

int mli_paged_prefill(const float* emb_table, const float* wpe, const int* inp, void* const* page_table,
                      const int* lengths, const int* new_item_indices, const void* wk, const void* wv, int n_batch,
                      int n_sequence, int emb_dim, int n_new_items, int elem_bf16, void* stream) {
    if (emb_table == nullptr || wpe == nullptr || inp == nullptr) return MLI_ERR_BAD_ARG;
    hipStream_t st = mli::as_stream(stream);
    if (elem_bf16 < MLI_ELEM_F32 || elem_bf16 > MLI_ELEM_FP8) return MLI_ERR_BAD_ARG;
    if (elem_bf16 == MLI_ELEM_FP8)
        return mli::launch_fill_paged_fp8_embed(emb_table, wpe, inp, reinterpret_cast<uint8_t* const*>(page_table),
                                                new_item_indices, lengths, static_cast<const mli_bf16*>(wk),
                                                static_cast<const mli_bf16*>(wv), n_batch, n_sequence, emb_dim, n_new_items, st);
    // The prologue form reads emb_table and wpe (fp32, 8 bytes per element) in EVERY column tile instead of the 2- or
    // 4-byte x element: worth one launch boundary and the x round trip while there are few column tiles, a loss beyond
    // (emb_dim 2048: 64 column tiles, 199 us against 154 + 14 us for encoder + fill; emb_dim 256: 8 tiles, a gain) -- so
    // wide models take the two launches.  Pages are bit-identical either way (tested).
    if (!mli::prefill_fuses(emb_dim)) {
        int rc = elem_bf16 ? mli_paged_attention_encoder_bf16(emb_table, wpe, inp, reinterpret_cast<mli_bf16* const*>(page_table),
                                                              lengths, new_item_indices, n_batch, n_sequence, emb_dim,
                                                              n_new_items, stream)
                           : mli_paged_attention_encoder(emb_table, wpe, inp, reinterpret_cast<float* const*>(page_table),
                                                         lengths, new_item_indices, n_batch, n_sequence, emb_dim, n_new_items,
                                                         stream);
        if (rc) return rc;
        return elem_bf16 ? mli_fill_new_k_v_cache_paged_bf16(reinterpret_cast<mli_bf16* const*>(page_table), new_item_indices,
                                                             lengths, static_cast<const mli_bf16*>(wk),
                                                             static_cast<const mli_bf16*>(wv), n_batch, n_sequence, emb_dim,
                                                             n_new_items, stream)
                         : mli_fill_new_k_v_cache_paged(reinterpret_cast<float* const*>(page_table), new_item_indices, lengths,
                                                        static_cast<const float*>(wk), static_cast<const float*>(wv), n_batch,
                                                        n_sequence, emb_dim, n_new_items, stream);
    }
    if (elem_bf16)
        return mli::launch_fill_paged_bf16_embed(emb_table, wpe, inp, reinterpret_cast<mli_bf16* const*>(page_table),
                                                 new_item_indices, lengths, static_cast<const mli_bf16*>(wk),
                                                 static_cast<const mli_bf16*>(wv), n_batch, n_sequence, emb_dim,
                                                 n_new_items, st);
    return mli::launch_fill_paged_embed(emb_table, wpe, inp, reinterpret_cast<float* const*>(page_table), new_item_indices,
                                        lengths, static_cast<const float*>(wk), static_cast<const float*>(wv), n_batch,
                                        n_sequence, emb_dim, n_new_items, st);
}

int mli_graph_begin_capture(void* stream) {
    if (stream == nullptr) return MLI_ERR_BAD_ARG;  // the legacy default stream cannot be captured
    return (int)hipStreamBeginCapture(mli::as_stream(stream), hipStreamCaptureModeThreadLocal);
}

int mli_graph_end_capture(void* stream, void** graph_exec_out) {
    if (stream == nullptr || graph_exec_out == nullptr) return MLI_ERR_BAD_ARG;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(mli::as_stream(stream), &graph);
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *graph_exec_out = exec;
    return 0;
}

int mli_stream_wait_stream(void* waiter, void* signaller) {
    // events are cheap to create and may be destroyed as soon as both calls have been issued: the runtime keeps what it
    // needs until the work has passed (eager), or has already turned it into a graph edge (capture)
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
    e = hipEventRecord(ev, mli::as_stream(signaller));
    if (e == hipSuccess) e = hipStreamWaitEvent(mli::as_stream(waiter), ev, 0);
    (void)hipEventDestroy(ev);
    return (int)e;
}

int mli_graph_launch(void* graph_exec, void* stream) {
    if (graph_exec == nullptr) return MLI_ERR_BAD_ARG;
    return (int)hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), mli::as_stream(stream));
}

int mli_graph_destroy(void* graph_exec) {
    if (graph_exec == nullptr) return 0;
    return (int)hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec));
}

int mli_paged_attention_bf16(mli_bf16* const* page_table, const int* lengths, const mli_bf16* wk, const mli_bf16* wq,
                             const mli_bf16* wv, const int* new_batch_idx, float* q_output, float* qkt_output,
                             float* attention_result, int n_batch, int n_sequence, int emb_dim, int n_new_items,
                             void* workspace, size_t workspace_bytes, void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
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
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
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
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
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
