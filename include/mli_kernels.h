/*
 * mli_kernels.h -- C ABI of libmli_hip.so, the MI355X (gfx950) implementation of the
 * single-block self-attention decode path of xyg-coder/min_llm_inference.
 *
 * This header is the drop-in boundary.  The reference has no FFI layer: its boundary is
 * the C++ `launch_*` free functions declared in the headers under include/kernels/ (reference paths
 * below are relative to the reference repository root).  Every entry point here is the
 * POD form of exactly one of those functions; the C++ adapters with the reference's own
 * signatures live in min_llm_inference_amd/host/src/launchers.cpp and only unpack
 * Tensor shapes before calling into this ABI.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - every buffer is caller-owned and pre-allocated; no entry point allocates, frees
 *     or synchronises (all are legal inside hipGraph capture);
 *   - `stream` is a hipStream_t passed as void*; NULL = the legacy default stream,
 *     which is what the reference launches on;
 *   - return value: 0 on success, a positive hipError_t when a launch failed, or
 *     MLI_ERR_BAD_ARG when a shape precondition the reference asserts on is violated;
 *   - indices are int32 as in the reference, offsets are computed in 64 bit;
 *   - page layout (reference include/utils.h:32-60): a page block holds
 *     PAGE_BLOCK_SIZE=16 tokens, float offset inside a block =
 *     (s % 16) * 3 * emb_dim + seg * emb_dim + d, seg 0 = input embedding, 1 = K, 2 = V;
 *     page table entry index = b * (n_sequence / 16) + s / 16.
 */
#ifndef MLI_KERNELS_H
#define MLI_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLI_PAGE_BLOCK_SIZE 16
#define MLI_EMPTY_ROW_TOKEN_ID (-1)
#define MLI_EOF_TOKEN_ID 1023
#define MLI_ERR_BAD_ARG (-22)
#define MLI_ERR_WORKSPACE (-12)

/* ABI version; bumped whenever a signature below changes. */
int mli_abi_version(void);

/* Page / weight element types of the lean entry points (their `elem` argument; the reference is fp32 only):
 *   MLI_ELEM_F32   float pages and weights -- the reference's types
 *   MLI_ELEM_BF16  EXTENSION: bfloat16 pages and weights (BASELINE config 4)
 *   MLI_ELEM_FP8   EXTENSION, opt-in: OCP e4m3 pages (x, K and V segments; one byte per element under the same layout
 *                  rule), bfloat16 weights; q, scores, softmax and every accumulation stay fp32
 * mli_elem_supported: 1 when this build implements the element type, 0 otherwise. */
#define MLI_ELEM_F32 0
#define MLI_ELEM_BF16 1
#define MLI_ELEM_FP8 2
int mli_elem_supported(int elem);

/* Bytes of device scratch the split-sequence kernels need for a problem of this size
 * (softmax_v / softmax_v_paged / paged_attention / inference_self_attention): per-chunk softmax statistics,
 * the split-sequence partial sums, and -- in a fixed 64 KiB region at the front, so that calls of different shapes can
 * share one buffer -- one arrival counter per batch row for the lean single-pass scan.  Never 0 for a valid shape.
 * The owner zero-fills that region ONCE after allocating the buffer (mli_attention_workspace_init); every entry point
 * leaves the counters zero on return, so nothing is re-initialised per call (nor under graph replay).  One workspace
 * serves one stream at a time. */
size_t mli_attention_workspace_bytes(int n_batch, int n_sequence, int dim);
/* Zero-fills the arrival counters at the front of a freshly allocated workspace (the first 64 KiB; a buffer that was
 * allocated zero-filled needs no call).  Stream-ordered; once per allocation, not per call. */
int mli_attention_workspace_init(void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Contiguous ("naive") KV-cache path.
 *   inp_embedding [n_batch, n_sequence, input_dim]   kt_cache [n_batch, output_dim, n_sequence]
 *   v_cache       [n_batch, n_sequence, output_dim]  wk/wq/wv [input_dim, output_dim]
 * ---------------------------------------------------------------------------------- */

/* replaces launch_fill_new_kt_v_cache  (include/kernels/self_attention_inference_optimized.h:5-8,
 * src/kernels/self_attention_inference_optimized.cu:303-323).  No-op when n_new_items == 0. */
int mli_fill_new_kt_v_cache(const float* inp_embedding, const int* new_batch_idx, const int* lengths,
                            const float* wk, const float* wv, float* kt_cache, float* v_cache,
                            int n_batch, int n_sequence, int input_dim, int output_dim,
                            int n_new_items, void* stream);

/* replaces launch_get_latest_kt_q_v  (…optimized.h:11-15, …optimized.cu:325-343).
 * Rows with lengths[b]==0 are left untouched (q_output included). */
int mli_get_latest_kt_q_v(const float* inp_embedding, const int* lengths,
                          const float* wk, const float* wq, const float* wv,
                          float* kt_cache, float* v_cache, float* q_output,
                          int n_batch, int n_sequence, int input_dim, int output_dim, void* stream);

/* replaces launch_qkt  (…optimized.h:17-19, …optimized.cu:345-358).
 * qkt_output[b, s] for s >= lengths[b] is not written. */
int mli_qkt(const float* q_output, const float* kt_cache, const int* lengths, float* qkt_output,
            int n_batch, int n_sequence, int dim, void* stream);

/* replaces launch_softmax_in_place_with_lengths  (…optimized.h:21-22, …optimized.cu:360-368).
 * Requires n_sequence % 4 == 0 (reference device assert, …optimized.cu:195).  Writes the whole
 * row: probabilities for s < lengths[b], 0 for the tail. */
int mli_softmax_in_place_with_lengths(float* qkt_output, const int* lengths,
                                      int n_batch, int n_sequence, void* stream);

/* replaces launch_softmax_v  (…optimized.h:24-26, …optimized.cu:370-383).
 * workspace may be NULL only when n_sequence <= 64 (single chunk: nothing is staged). */
int mli_softmax_v(const float* softmax_result, const float* v_cache, const int* lengths,
                  float* attention_result, int n_batch, int n_sequence, int output_dim,
                  void* workspace, size_t workspace_bytes, void* stream);

/* replaces inference_self_attention  (…optimized.h:28-48, …optimized.cu:282-301):
 * fill -> latest -> qkt -> softmax -> softmax_v on one stream. */
int mli_inference_self_attention(const float* inp_embedding, const int* lengths,
                                 const float* wk, const float* wq, const float* wv,
                                 const int* new_batch_idx, float* kt_cache, float* v_cache,
                                 float* q_output, float* qkt_output, float* attention_result,
                                 int n_batch, int n_sequence, int input_dim, int output_dim,
                                 int n_new_items, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Paged KV-cache path.  page_table is float*[n_batch][n_sequence/16] of device pointers.
 * Requires n_sequence % 16 == 0 and emb_dim % 4 == 0 (reference asserts,
 * src/kernels/paged_attention.cu:105-107, src/kernels/paged_attention_cublas.cu:212).
 * ---------------------------------------------------------------------------------- */

/* replaces launch_fill_new_k_v_cache_paged_attention (include/kernels/paged_attention.h:28-30,
 * src/kernels/paged_attention.cu:96-115) AND launch_fill_new_k_v_cache_paged_attention_warp_tiling
 * (paged_attention.h:65-67, src/kernels/paged_attention_cublas.cu:225-246): one MFMA kernel. */
int mli_fill_new_k_v_cache_paged(float* const* page_table, const int* new_batch_idx, const int* lengths,
                                 const float* wk, const float* wv,
                                 int n_batch, int n_sequence, int emb_dim, int n_new_items, void* stream);

/* replaces launch_get_latest_k_q_v_paged_attention (paged_attention.h:32-35, paged_attention.cu:188-199)
 * AND launch_get_latest_k_q_v_paged_attention_cublas (paged_attention.h:57-63,
 * paged_attention_cublas.cu:76-99: gather + 3 cublasSgemm + scatter) as one gather-GEMM-scatter kernel. */
int mli_get_latest_k_q_v_paged(float* const* page_table, const int* lengths,
                               const float* wk, const float* wq, const float* wv, float* q_output,
                               int n_batch, int n_sequence, int emb_dim, void* stream);

/* replaces launch_qkt_paged_attention (paged_attention.h:37-39, paged_attention.cu:270-280). */
int mli_qkt_paged(const float* q_output, const float* const* page_table, const int* lengths,
                  float* qkt_output, int n_batch, int n_sequence, int emb_dim, void* stream);

/* replaces launch_softmax_v_paged_attention (paged_attention.h:41-43, paged_attention.cu:333-345). */
int mli_softmax_v_paged(const float* softmax_result, const float* const* page_table, const int* lengths,
                        float* attention_result, int n_batch, int n_sequence, int emb_dim,
                        void* workspace, size_t workspace_bytes, void* stream);

/* replaces paged_attention (paged_attention.h:17-25, paged_attention.cu:358-377) and
 * paged_attention_with_cublas (paged_attention.h:46-54, paged_attention_cublas.cu:260-280).
 * On return q_output, qkt_output (probabilities, zero tail) and attention_result hold what the
 * reference's five launches leave there. */
int mli_paged_attention(float* const* page_table, const int* lengths,
                        const float* wk, const float* wq, const float* wv, const int* new_batch_idx,
                        float* q_output, float* qkt_output, float* attention_result,
                        int n_batch, int n_sequence, int emb_dim, int n_new_items,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * bfloat16 paged path (BASELINE.json config 4).  EXTENSION: the reference is fp32 only, so these entry
 * points have no reference counterpart; they are the fp32 paged entry points above with 16-bit pages
 * (same layout rule, elements are bf16) and bf16 weights.  q_output, qkt_output, attention_result and
 * every accumulation stay fp32.  Requires emb_dim % 8 == 0.  mli_bf16 = raw bfloat16 bits.
 * ---------------------------------------------------------------------------------- */
typedef uint16_t mli_bf16;

int mli_fill_new_k_v_cache_paged_bf16(mli_bf16* const* page_table, const int* new_batch_idx, const int* lengths,
                                      const mli_bf16* wk, const mli_bf16* wv,
                                      int n_batch, int n_sequence, int emb_dim, int n_new_items, void* stream);

int mli_get_latest_k_q_v_paged_bf16(mli_bf16* const* page_table, const int* lengths,
                                    const mli_bf16* wk, const mli_bf16* wq, const mli_bf16* wv, float* q_output,
                                    int n_batch, int n_sequence, int emb_dim, void* stream);

int mli_qkt_paged_bf16(const float* q_output, const mli_bf16* const* page_table, const int* lengths,
                       float* qkt_output, int n_batch, int n_sequence, int emb_dim, void* stream);

int mli_softmax_v_paged_bf16(const float* softmax_result, const mli_bf16* const* page_table, const int* lengths,
                             float* attention_result, int n_batch, int n_sequence, int emb_dim,
                             void* workspace, size_t workspace_bytes, void* stream);

int mli_paged_attention_bf16(mli_bf16* const* page_table, const int* lengths,
                             const mli_bf16* wk, const mli_bf16* wq, const mli_bf16* wv, const int* new_batch_idx,
                             float* q_output, float* qkt_output, float* attention_result,
                             int n_batch, int n_sequence, int emb_dim, int n_new_items,
                             void* workspace, size_t workspace_bytes, void* stream);

/* fp32 embedding tables in, bf16 input embedding written into segment 0 of the page */
int mli_paged_attention_encoder_bf16(const float* emb_table, const float* wpe, const int* inp,
                                     mli_bf16* const* page_table, const int* lengths, const int* new_item_indices,
                                     int n_batch, int n_sequence, int emb_dim, int n_new_items, void* stream);

int mli_paged_decoder_multi_rounds_bf16(const float* batch_result, const float* emb_table, float* emb_score,
                                        const float* wpe_table, mli_bf16* const* page_table, int* lengths,
                                        int* decoder_result, int n_batch, int n_vocab, int n_sequence, int emb_dim,
                                        int n_decoder_results, int i_decoder, void* stream);

/* LEAN paged composition -- what PagedAttention*Layer::forward runs.  The reference's paged_attention also leaves the
 * softmax probabilities in its qkt_output scratch (paged_attention.h:17-25), which nothing downstream reads
 * (src/layers.cpp:84-100 passes attention_result on, qkt_output stays in the layer).  This form computes the same
 * attention_result (bit-identical to mli_paged_attention[_bf16]) without materialising scores or probabilities:
 * fill (n_new_items rows) -> latest -> single-pass scan whose chunk results are merged inside the scan launch by the
 * workgroup that completes a row.  elem_bf16 selects the page / weight element type (0 = float, 1 = mli_bf16). */
int mli_paged_attention_lean(void* const* page_table, const int* lengths,
                             const void* wk, const void* wq, const void* wv, const int* new_batch_idx,
                             float* q_output, float* attention_result,
                             int n_batch, int n_sequence, int emb_dim, int n_new_items, int elem_bf16,
                             void* workspace, size_t workspace_bytes, void* stream);

/* The decode projection of mli_paged_attention_lean on its own (q, k, v of every non-empty row's last token; k, v appended to
 * the page, q to q_output) for any page element type: elem = MLI_ELEM_*.  For fp32 / bf16 pages it is
 * mli_get_latest_k_q_v_paged[_bf16]; fp8 pages have no other entry point for it.  (bench.py times it apart from the scan.) */
int mli_get_latest_k_q_v_paged_lean(void* const* page_table, const int* lengths, const void* wk, const void* wq,
                                    const void* wv, float* q_output, int n_batch, int n_sequence, int emb_dim, int elem,
                                    void* stream);

/* LEAN contiguous composition -- what SelfAttentionLayer::forward runs: inference_self_attention
 * (self_attention_inference_optimized.h:22-25) without its qkt_output scratch (nothing downstream reads it,
 * src/layers.cpp:41-56).  fill (n_new_items rows) -> latest -> ONE scan launch: a workgroup scores 256 tokens of a row
 * from the K^T tile, keeps exp(score - chunk max) in LDS, accumulates it over the V tile, and the workgroup that
 * completes a row merges its chunks (attention_fused_naive.hip).  attention_result differs from
 * mli_inference_self_attention's by fp32 rounding of the merge only.  output_dim and n_sequence must be multiples of 4
 * and the caches 16-byte aligned (the scan reads them in 16-byte pieces; input_dim is free); otherwise MLI_ERR_BAD_ARG
 * after the projection has run (it is idempotent) and the caller takes mli_inference_self_attention. */
int mli_self_attention_lean(const float* inp_embedding, const int* lengths,
                            const float* wk, const float* wq, const float* wv, const int* new_batch_idx,
                            float* kt_cache, float* v_cache, float* q_output, float* attention_result,
                            int n_batch, int n_sequence, int input_dim, int output_dim, int n_new_items,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The scan launch of mli_self_attention_lean on its own: q_output from mli_get_latest_kt_q_v in, attention_result out. */
int mli_decode_scan_contiguous(const float* q_output, const float* kt_cache, const float* v_cache, const int* lengths,
                               float* attention_result, int n_batch, int n_sequence, int emb_dim,
                               void* workspace, size_t workspace_bytes, void* stream);

/* The single-pass scan the paged compositions run after the projection (scores + masked softmax + softmax.V in
 * one visit per page; what A10 -> A4 -> A11 of SURVEY 8(a) compute together).  Inputs: q_output from
 * mli_get_latest_k_q_v_paged[_bf16]; outputs: qkt_output (probabilities, zero tail) and attention_result.
 * elem_bf16 selects the page element type; phases: 1 = scan kernel only, 2 = combine kernel only, 3 = both
 * (1 and 2 exist so the two launches can be timed apart); + 4 = lean mode: qkt_output is neither read nor written
 * (may be NULL) and the scan merges every row's chunks itself, so 5 = 7 = the whole job in ONE launch (6 = nothing;
 * with mli_tune("scan_merge", 0) the merge stays a second, slimmer launch and 5 / 6 time the two apart).  Rows of up to two 16-byte lane loads (fp32 <= 512,
 * bf16 <= 1024) give every wave whole pages; wider rows (fp32 <= 2048, bf16 <= 4096) are split across the four
 * waves of a workgroup.  Returns MLI_ERR_BAD_ARG beyond that: use the separate entry points then. */
int mli_decode_scan_paged(const float* q_output, const void* const* page_table, const int* lengths,
                          float* qkt_output, float* attention_result, int n_batch, int n_sequence, int emb_dim,
                          int elem_bf16, int phases, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Encoder / decoder head (needed for InferenceModel::forward).
 * ---------------------------------------------------------------------------------- */

/* replaces launch_inference_optimized_encoder_kernel (include/kernels/encoder.h:16-19,
 * src/kernels/encoder.cu:80-92). */
int mli_inference_optimized_encoder(const float* emb_table, const float* wpe, const int* inp,
                                    float* inp_embedding, const int* lengths, const int* new_item_indices,
                                    int n_batch, int n_sequence, int emb_dim, int n_new_items, void* stream);

/* replaces launch_paged_attention_encoder_kernel (encoder.h:22-25, encoder.cu:134-147). */
int mli_paged_attention_encoder(const float* emb_table, const float* wpe, const int* inp,
                                float* const* page_table, const int* lengths, const int* new_item_indices,
                                int n_batch, int n_sequence, int emb_dim, int n_new_items, void* stream);

/* replaces launch_decoder (include/kernels/decoder.h:19-23, src/kernels/decoder.cu:94-111):
 * emb_score = batch_result . emb_table^T, per-row argmax, lengths update, next embedding write. */
int mli_decoder(const float* batch_result, const float* emb_table, float* emb_score, const float* wpe_table,
                float* inp_embedding, int* lengths, int* decoder_result,
                int n_batch, int n_vocab, int n_sequence, int emb_dim, void* stream);

/* replaces launch_paged_attention_decoder_multi_rounds and
 * launch_paged_attention_cublas_decoder_multi_rounds (decoder.h:26-37, decoder.cu:207-255). */
int mli_paged_decoder_multi_rounds(const float* batch_result, const float* emb_table, float* emb_score,
                                   const float* wpe_table, float* const* page_table, int* lengths,
                                   int* decoder_result, int n_batch, int n_vocab, int n_sequence, int emb_dim,
                                   int n_decoder_results, int i_decoder, void* stream);

/* Decoder head with the argmax as the logits GEMM's EPILOGUE: emb_score[n_batch, n_vocab] is
 * never materialised.  Every 64-column tile of the product leaves one (max, lowest index of the max) pair per row in
 * `scratch` (mli_decoder_scratch_bytes), a one-wave-per-row kernel picks the row's token from those pairs, updates the
 * length and writes the next input embedding.  Tokens, lengths and embeddings are identical to mli_decoder /
 * mli_paged_decoder_multi_rounds[_bf16] (same fp32 products, same argmax order: larger value, then lower index) --
 * those stay for callers that want emb_score (the reference's decoder tests compare it).  What *DecoderLayer::forward
 * runs.  Replaces launch_decoder / launch_paged_attention[_cublas]_decoder_multi_rounds (decoder.h:19-37). */
size_t mli_decoder_scratch_bytes(int n_batch, int n_vocab);
int mli_decoder_fused(const float* batch_result, const float* emb_table, const float* wpe_table,
                      float* inp_embedding, int* lengths, int* decoder_result,
                      int n_batch, int n_vocab, int n_sequence, int emb_dim,
                      void* scratch, size_t scratch_bytes, void* stream);
int mli_paged_decoder_fused(const float* batch_result, const float* emb_table, const float* wpe_table,
                            void* const* page_table, int* lengths, int* decoder_result,
                            int n_batch, int n_vocab, int n_sequence, int emb_dim,
                            int n_decoder_results, int i_decoder, int elem_bf16,
                            void* scratch, size_t scratch_bytes, void* stream);

/* PREFILL of the newly inserted rows in ONE launch: the encoder as the fill GEMM's prologue.  The A
 * tile rows are computed on the fly as emb_table[inp[b, s]] + wpe[s] -- nothing is read back from the input-embedding
 * segment -- multiplied by [Wk | Wv], and the workgroups of the first column tile also write those rows to segment 0
 * (the decode projection reads position L - 1 from there).  Pages / caches come out bit-identical to
 *   mli_paged_attention_encoder[_bf16] + mli_fill_new_k_v_cache_paged[_bf16]   (paged; elem_bf16 selects the element type)
 *   mli_inference_optimized_encoder + mli_fill_new_kt_v_cache                   (contiguous)
 * i.e. launch_paged_attention_encoder_kernel + launch_fill_new_k_v_cache_paged_attention[_warp_tiling]
 * (encoder.h:22-25, paged_attention.h:28-30,65-67) resp. launch_inference_optimized_encoder_kernel +
 * launch_fill_new_kt_v_cache (encoder.h:16-19, self_attention_inference_optimized.h:5-8).  No-op when n_new_items == 0.
 * Since round 3 the entry points pick the form by shape (mli_tune "prefill_fused"): the prologue form up to emb_dim 512, those
 * two launches beyond -- every column tile of the prologue form re-reads the fp32 embedding and position rows, which wide
 * models (many column tiles) pay more for than for one launch boundary (emb_dim 2048: 199 us against 154 + 14 us). */
int mli_paged_prefill(const float* emb_table, const float* wpe, const int* inp, void* const* page_table,
                      const int* lengths, const int* new_item_indices, const void* wk, const void* wv,
                      int n_batch, int n_sequence, int emb_dim, int n_new_items, int elem_bf16, void* stream);
int mli_prefill(const float* emb_table, const float* wpe, const int* inp, float* inp_embedding, const int* lengths,
                const int* new_item_indices, const float* wk, const float* wv, float* kt_cache, float* v_cache,
                int n_batch, int n_sequence, int input_dim, int output_dim, int n_new_items, void* stream);

/* One whole decode step of the continuous batch (n_new_items = 0) in ONE call: what *InferenceModel::forward does per
 * round once the new rows are prefilled (reference src/inference_model.cpp:26-30, 68-72) -- lean attention
 * (mli_paged_attention_lean / mli_self_attention_lean) followed by the fused decoder head.  For hosts that pay per
 * call (the Python test / bench front end); the C++ layers issue the same launches themselves.
 *   paged:      q_output [n_batch, emb_dim] is scratch; attention_result [n_batch, emb_dim] holds the attention output
 *   contiguous: qkt_output [n_batch, n_sequence] is scratch for the shapes mli_self_attention_lean does not cover
 * The workspace must have been initialised once (mli_attention_workspace_init). */
int mli_paged_decode_step(void* const* page_table, int* lengths, const void* wk, const void* wq, const void* wv,
                          const float* emb_table, const float* wpe_table, float* q_output, float* attention_result,
                          int* decoder_result, int n_batch, int n_sequence, int emb_dim, int n_vocab,
                          int n_decoder_results, int i_decoder, int elem_bf16,
                          void* workspace, size_t workspace_bytes, void* decoder_scratch, size_t decoder_scratch_bytes,
                          void* stream);
int mli_decode_step(float* inp_embedding, int* lengths, const float* wk, const float* wq, const float* wv,
                    const float* emb_table, const float* wpe_table, float* kt_cache, float* v_cache,
                    float* q_output, float* qkt_output, float* attention_result, int* decoder_result,
                    int n_batch, int n_sequence, int emb_dim, int n_vocab,
                    void* workspace, size_t workspace_bytes, void* decoder_scratch, size_t decoder_scratch_bytes,
                    void* stream);

/* ------------------------------------------------------------------------------------
 * hipGraph capture of a decode step.  No entry point above allocates or synchronises, so any sequence of them issued
 * on one (non-default) stream between begin and end becomes a graph; replaying it costs one host call instead of one
 * per kernel.  A decode step with n_new_items == 0 is replay-safe: its launches depend on device state (lengths,
 * page table, pages) only through pointers, never through host-side values.  What the engines' forward() replays
 * (reference loop: src/inference_model.cpp:56-81).
 * ---------------------------------------------------------------------------------- */
int mli_graph_begin_capture(void* stream);                      /* stream must not be the legacy default stream */
int mli_graph_end_capture(void* stream, void** graph_exec_out); /* ends the capture and instantiates it */
int mli_graph_launch(void* graph_exec, void* stream);
/* `waiter` does not run anything queued after this call before everything queued on `signaller` so far has run (an
 * event recorded on one stream and waited for on the other).  Inside a capture it adds a dependency edge -- and pulls
 * `waiter` into the capture --, which is how a captured step forks into parallel branches (e.g. two micro-batches of
 * the batch rows: one half's memory-bound scan beside the other half's latency-bound GEMMs) and joins again. */
int mli_stream_wait_stream(void* waiter, void* signaller);
int mli_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------------------------
 * Test / measurement support (not on the reference's product path).
 * ---------------------------------------------------------------------------------- */

/* replaces launch_clone_inp_embedding_k_v_cache (include/utils.h:101-103, src/kernels/utils.cu:230-239):
 * copies contiguous inp/kt/v tensors into the pages of every non-empty row, positions
 * 0..min(length, n_sequence-1) inclusive. */
int mli_clone_inp_embedding_k_v_cache(float* const* page_table, const float* inp_embedding,
                                      const float* kt_cache, const float* v_cache, const int* lengths,
                                      int n_batch, int n_sequence, int emb_dim, void* stream);

/* fp32 -> fp8 (OCP e4m3fn; round to nearest even, saturating at +-448, NaN kept) with the conversion the fp8 page kernels
 * use: n (a multiple of 4) values.  For building synthetic fp8 pages in tests and bench.py. */
int mli_f32_to_fp8(const float* src, uint8_t* dst, size_t n, void* stream);

/* Tuning / diagnostic knobs.  PER THREAD: a setting applies to the launches the calling thread issues afterwards, so a test
 * or a probe never changes what an engine on another thread runs (results are identical for every setting):
 *   "chunk_tokens"     0 = heuristic, else a power of two in [64, 1024]: tokens per workgroup of the
 *                      split-sequence kernels
 *   "nt_loads"         non-temporal hint on the K/V stream: 2 (default) = where the rows' K/V (n_batch * n_sequence *
 *                      emb_dim * 2 elements) exceeds 768 MiB, i.e. nothing of it survives in the 256 MiB Infinity Cache
 *                      until the next step; 1 = always, 0 = never (plain loads)
 *   "qkt_token_batch"  4 | 8 (default) | 16: K rows a wave keeps in flight per load batch
 *   "flash_decode"     1 (default) = the paged compositions run the single-pass fused scan (each page visited
 *                      once for K and V, online softmax) when emb_dim fits (fp32 <= 2048, bf16 <= 4096);
 *                      0 = separate q.K^T / softmax / softmax.V passes
 *   "scan_partial_last" 1 (default) = the single-pass scan runs full chunks first and every row's remainder behind
 *                      them, cut into pieces of "scan_tail_tokens" tokens (what is still running when the queue runs
 *                      dry is short), 0 = plain chunk order
 *   "scan_tail_tokens" 0 (default) = the remainder stays one piece, else a power of two in [64, chunk]: piece size
 *   "scan_stream"      (lean mode) 1 (default) = batches that fill the chip (n_batch * n_sequence >= 2^21, n_batch <= 2048,
 *                      n_sequence >= 256) are scanned in EQUAL PAGE SHARES: the pages of all rows form one sequence
 *                      that 2 x CUs workgroups split evenly, each streaming its share across row boundaries and merging
 *                      per row like "scan_merge" (attention_stream.hip); 0 = always one workgroup per (row, chunk).
 *                      Same result for the same lengths on every launch; against the chunked form the split points of
 *                      a row differ, i.e. the fp32 rounding of the merge (<= 1e-5 on attention_result)
 *   "scan_stream_min_tokens"  the n_batch * n_sequence threshold of "scan_stream" (tests lower it)
 *   "scan_stream_dynamic_pct" / "scan_stream_granule"  the last x per cent (default 4, at most 12; 0 = none) of the page
 *                      sequence are not part of the equal shares but handed out by a ticket counter in granules of that
 *                      many pages (default 64, 16..256) to the workgroups that finish their share first
 *   "scan_merge"       (lean mode) 1 (default) = the workgroup that completes a row merges its chunks inside the scan
 *                      launch, 0 = a separate combine launch; bit-identical results
 *   "scan_dynamic_items" 1 = the single-pass scan hands its (row, chunk) items out through a ticket counter (balances
 *                      the XCDs on ragged lengths; the counter reset costs what it gains), 0 (default) = by grid position
 *   "fused_softmax"    (separate-pass form only) 1 = the compositions fold the masked softmax into the qkt / softmax.V kernels, 0 = three
 *                      launches as the reference (qkt, softmax_in_place_with_lengths, softmax_v), -1 (default) =
 *                      fuse when n_batch * n_sequence <= 2^20 (launch-bound steps)
 *   "fill_compact"     1 (default) = the prefill GEMM runs over the flat list of (new row, token) pairs, 0 = one tile
 *                      grid per new row (the reference's decomposition); bit-identical results
 *   "latest_compact"   1 (default) = the decode projection multiplies a device-built list of the non-empty batch rows where
 *                      the reduction is >= 1024 long (building the list costs every workgroup ~2 us), 2 = wherever
 *                      possible, 0 = all rows (empty ones as zeros); identical results
 *   "gemm_deep_k"      1 (default) = the bf16 GEMM stages 128 k per tile for latency-bound shapes, 0 = 32 everywhere
 *   "naive_scan_fused" 1 (default) = mli_self_attention_lean runs the single-launch contiguous scan, 0 = it returns
 *                      MLI_ERR_BAD_ARG (callers fall back to mli_inference_self_attention)
 *   "scan_row_order"   1 (default) = single-pass scans with one workgroup per row (short sequences) and more than 512 rows
 *                      hand the rows out longest first, 0 = in grid order (identical results)
 *   "gemm_split"       1 (default) = the fp32 GEMM with 64-row tiles (prefill, logits, projections of batches that do not
 *                      fill the chip with 128-row tiles) runs as 512-thread workgroups, four waves loading and four
 *                      multiplying, when the reduction is >= 256 long; 0 = one wave does both (identical results)
 *   "gemm_bf16_split"  1 (default) = the bf16 decode projection at emb_dim >= 1536 (any batch) or >= 1024 (from 320 rows),
 *                      emb_dim a multiple of 64, runs the loader-wave / MFMA-wave kernel (LDS-DMA loaders, 128 x 192 tiles, one
 *                      per CU at 1024 rows), 0 = the tiled kernels (identical results)
 *   "prefill_fused"    1 (default) = mli_[paged_]prefill runs the encoder as the fill GEMM's prologue up to emb_dim 512 and
 *                      encoder + fill as two launches beyond, 0 = always two launches, 2 = always the prologue form (fp8 pages
 *                      have the prologue form only); pages / caches are bit-identical either way
 *   "gemm_panel"       1 (default) = small fp32 products (emb_dim <= 512, fewer than 256 tiles of 64x64: the decode
 *                      projection and the logits of configs 2 / 3) run the latency-shaped kernel (32x32 tiles, the
 *                      whole K panel requested at once), 0 = always the tiled kernel, 2 = whenever the shape allows
 *                      (lets tests push other shapes through it); bit-identical results
 *   "gemm_tall_tiles"  1 (default) = 128x64 workgroup tiles for the decode projection / logits GEMM (fp32 and bf16)
 *                      when the grid still fills the chip, 0 = always 64x64, 2 = 128x64 whenever the mode allows it
 *                      (lets small test shapes exercise that kernel)
 *   "bf16_native_mfma" 1 (default) = v_mfma_f32_32x32x16_bf16 tile engine for the bf16 path, 0 = operands widened
 *                      to fp32 + fp32 MFMA (bit-identical to a sequential fp32 sum; differs from 1 only by the
 *                      rounding order of the fp32 accumulation) */
int mli_tune(const char* key, int value);

/* float4 device copy (kept for tools; a copy is not a ceiling for a read stream). */
int mli_stream_copy(const float* src, float* dst, size_t n_floats, void* stream);
/* Pure streaming read of n_floats (a multiple of 16384) with the scan's load shape -- 16-byte non-temporal lane loads,
 * 16 KiB per wave in flight, nothing written: the box's read-only HBM rate, which bench.py reports beside the scan's
 * (`roofline.measured_read_gbs`).  sink: >= 1 float of device memory, never written in practice. */
int mli_stream_read(const float* src, float* sink, size_t n_floats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLI_KERNELS_H */
