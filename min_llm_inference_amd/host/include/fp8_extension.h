// fp8 (OCP e4m3) KV pages for the C++ front end.  EXTENSION, opt-in ("page-pool layout v2 ... fp8 KV"):
// the reference is fp32 only, so nothing here has a reference counterpart; the classes follow the shape of the bf16
// extension (bf16_extension.h) so a caller switches by changing type names.  Pages hold one byte per element under the
// same layout rule (16 tokens x [x | K | V] x emb_dim: a page is 48 * emb_dim bytes, a quarter of the fp32 page), the
// weights are bf16, q / scores / softmax / every accumulation / attention_result / logits stay fp32.  Only the lean
// compositions exist for this element type (mli_paged_prefill, mli_paged_attention_lean, mli_paged_decoder_fused with
// elem = MLI_ELEM_FP8): there is no scores-materialising form to fall back to.
#pragma once

#include "bf16_extension.h"

// floats to ask MemoryBlockManager for so that one block holds 16 x 3 x emb_dim fp8 elements
inline size_t fp8_page_block_floats(size_t emb_dim) { return 16 * 3 * emb_dim / 4; }

class PagedAttentionFp8Layer : public NonCopyableNonClonable {
public:
    PagedAttentionFp8Layer(TensorBf16&& wk, TensorBf16&& wq, TensorBf16&& wv, size_t n_batch, size_t emb_dim,
                           size_t n_sequence);
    void forward(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_batch_idx,
                 TensorFloat& attention_result, int n_new_items);
    // encoder + K/V prefill of the new rows in one launch (mli_paged_prefill, elem = MLI_ELEM_FP8)
    void prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                 int n_new_items);

private:
    TensorBf16 wk_, wq_, wv_;
    TensorFloat q_output_;
    size_t n_sequence_;
};

class PagedAttentionFp8InferenceModel : public NonCopyableNonClonable {
public:
    PagedAttentionFp8InferenceModel(PagedAttentionFp8Layer&&, size_t n_batch, size_t n_sequence, size_t emb_dim,
                                    size_t n_vocab, int n_forward_rounds);
    void forward(const TensorInt& inp, TensorInt& lengths, const TensorInt& new_item_indices,
                 TensorInt& decoder_result, int n_new_items, const TensorFloat& emb_table,
                 const TensorFloat& pos_emb_table, TensorFloatPoint& page_table);

private:
    PagedAttentionFp8Layer attention_layer_;
    size_t n_batch_, n_sequence_, emb_dim_;
    TensorFloat attention_result_;
    TensorFloat decoder_scratch_;
    int n_forward_rounds_;
};

void start_paged_attention_fp8_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                MemoryBlockManager& memory_block_manager,
                                                PagedAttentionsManager& paged_attention_manager,
                                                PagedAttentionFp8InferenceModel& inference_model,
                                                size_t n_batch_size, size_t n_sequence, int n_forward_rounds);
