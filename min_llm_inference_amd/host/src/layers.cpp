#include "layers.h"

#include <utility>

#include "kernels/decoder.h"
#include "kernels/encoder.h"
#include "kernels/paged_attention.h"
#include "kernels/self_attention_inference_optimized.h"
#include "runtime.h"

namespace {
TensorFloat device_tensor(std::initializer_list<size_t> shape) {
    return TensorFloat(std::vector<size_t>(shape), DeviceType::DEVICE);
}
}  // namespace

SelfAttentionLayer::SelfAttentionLayer(TensorFloat&& wk, TensorFloat&& wq, TensorFloat&& wv, size_t n_batch,
                                       size_t input_dim, size_t n_sequence)
    : wk_(std::move(wk)), wq_(std::move(wq)), wv_(std::move(wv)),
      kt_cache_(device_tensor({n_batch, input_dim, n_sequence})),
      v_cache_(device_tensor({n_batch, n_sequence, input_dim})),
      q_output_(device_tensor({n_batch, input_dim})),
      qkt_output_(device_tensor({n_batch, n_sequence})) {}

void SelfAttentionLayer::forward(const TensorFloat& inp_embedding, const TensorInt& lengths,
                                 const TensorInt& new_batch_idx, TensorFloat& attention_result, int n_new_items) {
    if (mli::runtime::lean_layers())
        inference_self_attention_lean(inp_embedding, lengths, wk_, wq_, wv_, new_batch_idx, kt_cache_, v_cache_, q_output_,
                                      qkt_output_, attention_result, n_new_items);
    else
        inference_self_attention(inp_embedding, lengths, wk_, wq_, wv_, new_batch_idx, kt_cache_, v_cache_, q_output_,
                                 qkt_output_, attention_result, n_new_items);
}

void SelfAttentionLayer::prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                                 TensorFloat& inp_embedding, const TensorInt& lengths,
                                 const TensorInt& new_item_indices, int n_new_items) {
    launch_prefill(emb_table, pos_emb, inp, inp_embedding, lengths, new_item_indices, wk_, wv_, kt_cache_, v_cache_,
                   n_new_items);
}

PagedAttentionLayer::PagedAttentionLayer(TensorFloat&& wk, TensorFloat&& wq, TensorFloat&& wv, size_t n_batch,
                                         size_t emb_dim, size_t n_sequence)
    : wk_(std::move(wk)), wq_(std::move(wq)), wv_(std::move(wv)),
      q_output_(device_tensor({n_batch, emb_dim})),
      qkt_output_(device_tensor({n_batch, n_sequence})) {}

void PagedAttentionLayer::forward(TensorFloatPoint& page_table, const TensorInt& lengths,
                                  const TensorInt& new_batch_idx, TensorFloat& attention_result, int n_new_items) {
    const int n_sequence = static_cast<int>(qkt_output_.shape()[1]);
    if (mli::runtime::lean_layers())
        paged_attention_lean(page_table, lengths, wk_, wq_, wv_, new_batch_idx, q_output_, qkt_output_,
                             attention_result, n_new_items, n_sequence);
    else
        paged_attention(page_table, lengths, wk_, wq_, wv_, new_batch_idx, q_output_, qkt_output_, attention_result,
                        n_new_items, n_sequence);
}

void PagedAttentionLayer::prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                                  TensorFloatPoint& page_table, const TensorInt& lengths,
                                  const TensorInt& new_item_indices, int n_new_items) {
    launch_paged_prefill(emb_table, pos_emb, inp, page_table, lengths, new_item_indices, wk_, wv_, n_new_items);
}

void PagedAttentionCublasLayer::prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                                        TensorFloatPoint& page_table, const TensorInt& lengths,
                                        const TensorInt& new_item_indices, int n_new_items) {
    launch_paged_prefill(emb_table, pos_emb, inp, page_table, lengths, new_item_indices, wk_, wv_, n_new_items);
}

PagedAttentionCublasLayer::PagedAttentionCublasLayer(TensorFloat&& wk, TensorFloat&& wq, TensorFloat&& wv,
                                                     size_t n_batch, size_t emb_dim, size_t n_sequence)
    : wk_(std::move(wk)), wq_(std::move(wq)), wv_(std::move(wv)),
      q_output_(device_tensor({n_batch, emb_dim})),
      qkt_output_(device_tensor({n_batch, n_sequence})),
      latest_emb_(device_tensor({n_batch, emb_dim})),
      temp_placeholder_(device_tensor({n_batch, emb_dim})) {}

void PagedAttentionCublasLayer::forward(TensorFloatPoint& page_table, const TensorInt& lengths,
                                        const TensorInt& new_batch_idx, TensorFloat& attention_result,
                                        int n_new_items, GemmHandle& handle) {
    const int n_sequence = static_cast<int>(qkt_output_.shape()[1]);
    if (mli::runtime::lean_layers())
        paged_attention_lean(page_table, lengths, wk_, wq_, wv_, new_batch_idx, q_output_, qkt_output_,
                             attention_result, n_new_items, n_sequence);
    else
        paged_attention_with_cublas(page_table, lengths, wk_, wq_, wv_, new_batch_idx, q_output_, qkt_output_,
                                    attention_result, latest_emb_, temp_placeholder_, n_new_items, n_sequence, handle);
}

void EncoderLayer::forward(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                           TensorFloat& inp_embedding, const TensorInt& lengths, const TensorInt& new_item_indices,
                           int n_new_items) {
    const auto& s = inp_embedding.shape();
    launch_inference_optimized_encoder_kernel(emb_table.data(), pos_emb.data(), inp.data(), inp_embedding.data(),
                                              lengths.data(), new_item_indices.data(), static_cast<int>(s[0]),
                                              static_cast<int>(s[1]), static_cast<int>(s[2]), n_new_items);
}

void PagedEncoderLayer::forward(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                                TensorFloatPoint& page_table, const TensorInt& lengths,
                                const TensorInt& new_item_indices, int n_new_items) {
    launch_paged_attention_encoder_kernel(emb_table.data(), pos_emb.data(), inp.data(), page_table.data(),
                                          lengths.data(), new_item_indices.data(), static_cast<int>(inp.shape()[0]),
                                          static_cast<int>(inp.shape()[1]), static_cast<int>(emb_table.shape()[1]),
                                          n_new_items);
}

DecoderLayer::DecoderLayer(size_t n_batch, size_t n_vocab) : emb_score_(device_tensor({n_batch, n_vocab})) {}

void DecoderLayer::forward(const TensorFloat& batch_result, const TensorFloat& emb_table,
                           const TensorFloat& wpe_table, TensorFloat& inp_embedding, TensorInt& lengths,
                           TensorInt& decoder_result) {
    if (mli::runtime::lean_layers())
        launch_decoder_fused(batch_result, emb_table, emb_score_, wpe_table, inp_embedding, lengths, decoder_result);
    else
        launch_decoder(batch_result, emb_table, emb_score_, wpe_table, inp_embedding, lengths, decoder_result);
}

PagedDecoderLayer::PagedDecoderLayer(size_t n_batch, size_t n_vocab)
    : emb_score_(device_tensor({n_batch, n_vocab})) {}

void PagedDecoderLayer::forward(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                const TensorFloat& wpe_table, TensorFloatPoint& page_table, TensorInt& lengths,
                                TensorInt& decoder_result, int i_decoder_round) {
    if (mli::runtime::lean_layers())
        launch_paged_attention_decoder_fused(batch_result, emb_table, emb_score_, wpe_table, page_table, lengths,
                                             decoder_result, i_decoder_round);
    else
        launch_paged_attention_decoder_multi_rounds(batch_result, emb_table, emb_score_, wpe_table, page_table,
                                                    lengths, decoder_result, i_decoder_round);
}

PagedCublasDecoderLayer::PagedCublasDecoderLayer(size_t n_batch, size_t n_vocab)
    : emb_score_(device_tensor({n_batch, n_vocab})) {}

void PagedCublasDecoderLayer::forward(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                      const TensorFloat& wpe_table, TensorFloatPoint& page_table,
                                      TensorInt& lengths, TensorInt& decoder_result, int i_decoder_round,
                                      GemmHandle& handle) {
    if (mli::runtime::lean_layers())
        launch_paged_attention_decoder_fused(batch_result, emb_table, emb_score_, wpe_table, page_table, lengths,
                                             decoder_result, i_decoder_round);
    else
        launch_paged_attention_cublas_decoder_multi_rounds(batch_result, emb_table, emb_score_, wpe_table, page_table,
                                                           lengths, decoder_result, i_decoder_round, handle);
}
