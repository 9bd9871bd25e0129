// bfloat16 paged path for the C++ front end (BASELINE config 4).  EXTENSION: the reference is fp32 only, so nothing
// in this header has a reference counterpart; it follows the shape of the fp32 paged classes
// (PagedAttentionCublasLayer / PagedAttentionCublasInferenceModel / start_paged_attention_cublas_inference_engine,
// reference include/layers.h:46-66, include/inference_model.h:52-74, include/inferencer.h:26-32) so a caller
// switches by changing type names.  Pages hold bf16 elements under the same layout rule (16 tokens x
// [x | K | V] x emb_dim), weights are bf16, q / scores / accumulation / attention_result / logits stay fp32.
// The page table stays a TensorFloatPoint: its entries are opaque block addresses and the pool is sized in bytes.
#pragma once

#include <cstdint>
#include <vector>

#include "inference_model.h"
#include "item_storage.h"
#include "paged_item_storage.h"

typedef Tensor<uint16_t> TensorBf16;  // raw bfloat16 bits

// round-to-nearest-even fp32 -> bf16 of a HOST fp32 tensor, uploaded to the device
TensorBf16 make_device_bf16(const float* host_values, std::vector<size_t> shape);

// floats to ask MemoryBlockManager for so that one block holds 16 x 3 x emb_dim bf16 elements
inline size_t bf16_page_block_floats(size_t emb_dim) { return 16 * 3 * emb_dim / 2; }

void paged_attention_bf16(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorBf16& wk,
                          const TensorBf16& wq, const TensorBf16& wv, const TensorInt& new_batch_idx,
                          TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result,
                          int n_new_items, int n_sequence);

void launch_paged_attention_encoder_kernel_bf16(const float* emb_table, const float* wpe, const int* inp,
                                                float** page_table, const int* lengths,
                                                const int* new_item_indices, int batch_size, int n_sequence,
                                                int embedding_dim, int n_new_items);

void launch_paged_attention_decoder_multi_rounds_bf16(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                                      TensorFloat& emb_score, const TensorFloat& wpe_table,
                                                      TensorFloatPoint& page_table, TensorInt& lengths,
                                                      TensorInt& decoder_result, int i_decoder);

class PagedAttentionBf16Layer : public NonCopyableNonClonable {
public:
    PagedAttentionBf16Layer(TensorBf16&& wk, TensorBf16&& wq, TensorBf16&& wv, size_t n_batch, size_t emb_dim,
                            size_t n_sequence);
    void forward(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_batch_idx,
                 TensorFloat& attention_result, int n_new_items);
    // encoder + K/V prefill of the new rows in one launch (mli_paged_prefill, elem_bf16 = 1)
    void prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                 int n_new_items);

private:
    TensorBf16 wk_, wq_, wv_;
    TensorFloat q_output_;
    TensorFloat qkt_output_;
};

class PagedAttentionBf16InferenceModel : public NonCopyableNonClonable {
public:
    PagedAttentionBf16InferenceModel(PagedAttentionBf16Layer&&, size_t n_batch, size_t n_sequence, size_t emb_dim,
                                     size_t n_vocab, int n_forward_rounds);
    void forward(const TensorInt& inp, TensorInt& lengths, const TensorInt& new_item_indices,
                 TensorInt& decoder_result, int n_new_items, const TensorFloat& emb_table,
                 const TensorFloat& pos_emb_table, TensorFloatPoint& page_table);

private:
    PagedAttentionBf16Layer attention_layer_;
    size_t n_batch_, n_sequence_, emb_dim_;
    TensorFloat attention_result_;
    TensorFloat emb_score_;
    int n_forward_rounds_;
};

void start_paged_attention_bf16_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                 ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                 MemoryBlockManager& memory_block_manager,
                                                 PagedAttentionsManager& paged_attention_manager,
                                                 PagedAttentionBf16InferenceModel& inference_model,
                                                 size_t n_batch_size, size_t n_sequence, int n_forward_rounds);
