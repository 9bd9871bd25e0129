#include "bf16_extension.h"

#include <cstring>
#include <utility>

#include "mli_kernels.h"
#include "runtime.h"
#include "utils.h"

namespace {

uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}

mli_bf16* const* bf16_pages(const TensorFloatPoint& t) { return reinterpret_cast<mli_bf16* const*>(t.data()); }

}  // namespace

TensorBf16 make_device_bf16(const float* host_values, std::vector<size_t> shape) {
    TensorBf16 staging(shape, DeviceType::HOST);
    const size_t n = staging.get_total_size();
    for (size_t i = 0; i < n; ++i) staging.data()[i] = f32_to_bf16_rne(host_values[i]);
    TensorBf16 device(shape, DeviceType::DEVICE);
    device.copy_from(staging);
    return device;
}

void paged_attention_bf16(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorBf16& wk,
                          const TensorBf16& wq, const TensorBf16& wv, const TensorInt& new_batch_idx,
                          TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result,
                          int n_new_items, int n_sequence) {
    const int B = (int)page_table.shape()[0], D = (int)wk.shape()[0];
    const mli::runtime::Scratch ws = mli::runtime::attention_scratch(B, n_sequence, D);
    HIP_CHECK(mli_paged_attention_bf16(bf16_pages(page_table), lengths.data(), wk.data(), wq.data(), wv.data(),
                                       new_batch_idx.data(), q_output.data(), qkt_output.data(),
                                       attention_result.data(), B, n_sequence, D, n_new_items,
                                       ws.ptr, ws.bytes, mli::runtime::compute_stream()));
}

void launch_paged_attention_encoder_kernel_bf16(const float* emb_table, const float* wpe, const int* inp,
                                                float** page_table, const int* lengths,
                                                const int* new_item_indices, int batch_size, int n_sequence,
                                                int embedding_dim, int n_new_items) {
    HIP_CHECK(mli_paged_attention_encoder_bf16(emb_table, wpe, inp, reinterpret_cast<mli_bf16* const*>(page_table),
                                               lengths, new_item_indices, batch_size, n_sequence, embedding_dim,
                                               n_new_items, mli::runtime::compute_stream()));
}

void launch_paged_attention_decoder_multi_rounds_bf16(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                                      TensorFloat& emb_score, const TensorFloat& wpe_table,
                                                      TensorFloatPoint& page_table, TensorInt& lengths,
                                                      TensorInt& decoder_result, int i_decoder) {
    const int n_results = decoder_result.shape().size() == 2 ? (int)decoder_result.shape()[1] : 1;
    HIP_CHECK(mli_paged_decoder_multi_rounds_bf16(batch_result.data(), emb_table.data(), emb_score.data(),
                                                  wpe_table.data(), bf16_pages(page_table), lengths.data(),
                                                  decoder_result.data(), (int)batch_result.shape()[0],
                                                  (int)emb_table.shape()[0], (int)wpe_table.shape()[0],
                                                  (int)batch_result.shape()[1], n_results, i_decoder,
                                                  mli::runtime::compute_stream()));
}

PagedAttentionBf16Layer::PagedAttentionBf16Layer(TensorBf16&& wk, TensorBf16&& wq, TensorBf16&& wv, size_t n_batch,
                                                 size_t emb_dim, size_t n_sequence)
    : wk_(std::move(wk)), wq_(std::move(wq)), wv_(std::move(wv)),
      q_output_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE),
      qkt_output_(std::vector<size_t>{n_batch, n_sequence}, DeviceType::DEVICE) {}

void PagedAttentionBf16Layer::forward(TensorFloatPoint& page_table, const TensorInt& lengths,
                                      const TensorInt& new_batch_idx, TensorFloat& attention_result,
                                      int n_new_items) {
    const int n_sequence = static_cast<int>(qkt_output_.shape()[1]);
    if (mli::runtime::lean_layers()) {
        const int B = (int)page_table.shape()[0], D = (int)wk_.shape()[0];
        const mli::runtime::Scratch ws = mli::runtime::attention_scratch(B, n_sequence, D);
        const int rc = mli_paged_attention_lean(reinterpret_cast<void* const*>(page_table.data()), lengths.data(),
                                                wk_.data(), wq_.data(), wv_.data(), new_batch_idx.data(),
                                                q_output_.data(), attention_result.data(), B, n_sequence, D,
                                                n_new_items, /*elem_bf16=*/1, ws.ptr, ws.bytes,
                                                mli::runtime::compute_stream());
        if (rc != MLI_ERR_BAD_ARG || D <= 4096) {  // rows wider than the single-pass kernel covers: fall through
            HIP_CHECK(rc);
            return;
        }
    }
    paged_attention_bf16(page_table, lengths, wk_, wq_, wv_, new_batch_idx, q_output_, qkt_output_, attention_result,
                         n_new_items, n_sequence);
}

void PagedAttentionBf16Layer::prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                                      TensorFloatPoint& page_table, const TensorInt& lengths,
                                      const TensorInt& new_item_indices, int n_new_items) {
    if (n_new_items == 0) return;
    HIP_CHECK(mli_paged_prefill(emb_table.data(), pos_emb.data(), inp.data(),
                                reinterpret_cast<void* const*>(page_table.data()), lengths.data(), new_item_indices.data(),
                                wk_.data(), wv_.data(), (int)inp.shape()[0], (int)inp.shape()[1],
                                (int)emb_table.shape()[1], n_new_items, /*elem_bf16=*/1, mli::runtime::compute_stream()));
}

PagedAttentionBf16InferenceModel::PagedAttentionBf16InferenceModel(PagedAttentionBf16Layer&& attention_layer,
                                                                   size_t n_batch, size_t n_sequence, size_t emb_dim,
                                                                   size_t n_vocab, int n_forward_rounds)
    : attention_layer_(std::move(attention_layer)), n_batch_(n_batch), n_sequence_(n_sequence), emb_dim_(emb_dim),
      attention_result_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE),
      emb_score_(std::vector<size_t>{n_batch, n_vocab}, DeviceType::DEVICE), n_forward_rounds_(n_forward_rounds) {}

void PagedAttentionBf16InferenceModel::forward(const TensorInt& inp, TensorInt& lengths,
                                               const TensorInt& new_item_indices, TensorInt& decoder_result,
                                               int n_new_items, const TensorFloat& emb_table,
                                               const TensorFloat& pos_emb_table, TensorFloatPoint& page_table) {
    for (int round = 0; round < n_forward_rounds_; ++round) {
        const int fresh = round == 0 ? n_new_items : 0;  // later rounds only decode
        if (mli::runtime::lean_layers()) {
            attention_layer_.prefill(emb_table, pos_emb_table, inp, page_table, lengths, new_item_indices, fresh);
            attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, 0);
        } else {
            launch_paged_attention_encoder_kernel_bf16(emb_table.data(), pos_emb_table.data(), inp.data(),
                                                       page_table.data(), lengths.data(), new_item_indices.data(),
                                                       (int)n_batch_, (int)n_sequence_, (int)emb_dim_, fresh);
            attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, fresh);
        }
        if (mli::runtime::lean_layers()) {
            HIP_CHECK(mli_paged_decoder_fused(attention_result_.data(), emb_table.data(), pos_emb_table.data(),
                                              reinterpret_cast<void* const*>(page_table.data()), lengths.data(),
                                              decoder_result.data(), (int)n_batch_, (int)emb_table.shape()[0],
                                              (int)n_sequence_, (int)emb_dim_, n_forward_rounds_, round, /*elem_bf16=*/1,
                                              emb_score_.data(), emb_score_.get_total_size() * sizeof(float),
                                              mli::runtime::compute_stream()));
        } else {
            launch_paged_attention_decoder_multi_rounds_bf16(attention_result_, emb_table, emb_score_, pos_emb_table,
                                                             page_table, lengths, decoder_result, round);
        }
    }
}
