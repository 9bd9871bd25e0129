#include "fp8_extension.h"

#include <utility>

#include "mli_kernels.h"
#include "runtime.h"
#include "utils.h"

PagedAttentionFp8Layer::PagedAttentionFp8Layer(TensorBf16&& wk, TensorBf16&& wq, TensorBf16&& wv, size_t n_batch,
                                               size_t emb_dim, size_t n_sequence)
    : wk_(std::move(wk)), wq_(std::move(wq)), wv_(std::move(wv)),
      q_output_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE), n_sequence_(n_sequence) {}

void PagedAttentionFp8Layer::forward(TensorFloatPoint& page_table, const TensorInt& lengths,
                                     const TensorInt& new_batch_idx, TensorFloat& attention_result, int n_new_items) {
    const int B = (int)page_table.shape()[0], D = (int)wk_.shape()[0], S = (int)n_sequence_;
    const mli::runtime::Scratch ws = mli::runtime::attention_scratch(B, S, D);
    HIP_CHECK(mli_paged_attention_lean(reinterpret_cast<void* const*>(page_table.data()), lengths.data(), wk_.data(),
                                       wq_.data(), wv_.data(), new_batch_idx.data(), q_output_.data(),
                                       attention_result.data(), B, S, D, n_new_items, MLI_ELEM_FP8, ws.ptr, ws.bytes,
                                       mli::runtime::compute_stream()));
}

void PagedAttentionFp8Layer::prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                                     TensorFloatPoint& page_table, const TensorInt& lengths,
                                     const TensorInt& new_item_indices, int n_new_items) {
    if (n_new_items == 0) return;
    HIP_CHECK(mli_paged_prefill(emb_table.data(), pos_emb.data(), inp.data(),
                                reinterpret_cast<void* const*>(page_table.data()), lengths.data(), new_item_indices.data(),
                                wk_.data(), wv_.data(), (int)inp.shape()[0], (int)inp.shape()[1],
                                (int)emb_table.shape()[1], n_new_items, MLI_ELEM_FP8, mli::runtime::compute_stream()));
}

PagedAttentionFp8InferenceModel::PagedAttentionFp8InferenceModel(PagedAttentionFp8Layer&& attention_layer, size_t n_batch,
                                                                 size_t n_sequence, size_t emb_dim, size_t n_vocab,
                                                                 int n_forward_rounds)
    : attention_layer_(std::move(attention_layer)), n_batch_(n_batch), n_sequence_(n_sequence), emb_dim_(emb_dim),
      attention_result_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE),
      decoder_scratch_(std::vector<size_t>{(mli_decoder_scratch_bytes((int)n_batch, (int)n_vocab) + 3) / 4}, DeviceType::DEVICE),
      n_forward_rounds_(n_forward_rounds) {}

void PagedAttentionFp8InferenceModel::forward(const TensorInt& inp, TensorInt& lengths, const TensorInt& new_item_indices,
                                              TensorInt& decoder_result, int n_new_items, const TensorFloat& emb_table,
                                              const TensorFloat& pos_emb_table, TensorFloatPoint& page_table) {
    for (int round = 0; round < n_forward_rounds_; ++round) {
        const int fresh = round == 0 ? n_new_items : 0;  // later rounds only decode
        attention_layer_.prefill(emb_table, pos_emb_table, inp, page_table, lengths, new_item_indices, fresh);
        attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, 0);
        HIP_CHECK(mli_paged_decoder_fused(attention_result_.data(), emb_table.data(), pos_emb_table.data(),
                                          reinterpret_cast<void* const*>(page_table.data()), lengths.data(),
                                          decoder_result.data(), (int)n_batch_, (int)emb_table.shape()[0],
                                          (int)n_sequence_, (int)emb_dim_, n_forward_rounds_, round, MLI_ELEM_FP8,
                                          decoder_scratch_.data(), decoder_scratch_.get_total_size() * sizeof(float),
                                          mli::runtime::compute_stream()));
    }
}
