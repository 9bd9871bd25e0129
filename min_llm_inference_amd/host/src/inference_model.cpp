#include "inference_model.h"

#include <utility>

#include "runtime.h"

InferenceModel::InferenceModel(SelfAttentionLayer&& attention_layer, EncoderLayer&& encoder_layer,
                               DecoderLayer&& decoder_layer, size_t n_batch, size_t n_sequence, size_t emb_dim)
    : attention_layer_(std::move(attention_layer)), encoder_layer_(std::move(encoder_layer)),
      decoder_layer_(std::move(decoder_layer)), n_batch_(n_batch), n_sequence_(n_sequence), emb_dim_(emb_dim),
      inp_embedding_(std::vector<size_t>{n_batch, n_sequence, emb_dim}, DeviceType::DEVICE),
      attention_result_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE) {}

void InferenceModel::forward(const TensorInt& inp, TensorInt& lengths, const TensorInt& new_item_indices,
                             TensorInt& decoder_result, int n_new_items, const TensorFloat& emb_table,
                             const TensorFloat& pos_emb_table) {
    if (mli::runtime::lean_layers()) {
        // encoder + prefill of the new rows in one launch (the embedding lookup is the fill GEMM's prologue), then a pure
        // decode step; same inp_embedding / caches as the two launches below
        attention_layer_.prefill(emb_table, pos_emb_table, inp, inp_embedding_, lengths, new_item_indices, n_new_items);
        attention_layer_.forward(inp_embedding_, lengths, new_item_indices, attention_result_, 0);
    } else {
        encoder_layer_.forward(emb_table, pos_emb_table, inp, inp_embedding_, lengths, new_item_indices, n_new_items);
        attention_layer_.forward(inp_embedding_, lengths, new_item_indices, attention_result_, n_new_items);
    }
    decoder_layer_.forward(attention_result_, emb_table, pos_emb_table, inp_embedding_, lengths, decoder_result);
}

PagedAttentionInferenceModel::PagedAttentionInferenceModel(PagedAttentionLayer&& attention_layer,
                                                           PagedEncoderLayer&& encoder_layer,
                                                           PagedDecoderLayer&& decoder_layer, size_t n_batch,
                                                           size_t n_sequence, size_t emb_dim, int n_forward_rounds)
    : paged_attention_layer_(std::move(attention_layer)), paged_encoder_layer_(std::move(encoder_layer)),
      paged_decoder_layer_(std::move(decoder_layer)), n_batch_(n_batch), n_sequence_(n_sequence),
      emb_dim_(emb_dim), attention_result_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE),
      n_forward_rounds_(n_forward_rounds) {}

void PagedAttentionInferenceModel::forward(const TensorInt& inp, TensorInt& lengths,
                                           const TensorInt& new_item_indices, TensorInt& decoder_result,
                                           int n_new_items, const TensorFloat& emb_table,
                                           const TensorFloat& pos_emb_table, TensorFloatPoint& page_table) {
    auto rounds = [&](int n_new) {
        for (int round = 0; round < n_forward_rounds_; ++round) {
            const int fresh = round == 0 ? n_new : 0;  // later rounds only decode
            if (mli::runtime::lean_layers()) {
                paged_attention_layer_.prefill(emb_table, pos_emb_table, inp, page_table, lengths, new_item_indices, fresh);
                paged_attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, 0);
            } else {
                paged_encoder_layer_.forward(emb_table, pos_emb_table, inp, page_table, lengths, new_item_indices, fresh);
                paged_attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, fresh);
            }
            paged_decoder_layer_.forward(attention_result_, emb_table, pos_emb_table, page_table, lengths,
                                         decoder_result, round);
        }
    };
    if (n_new_items != 0) {
        rounds(n_new_items);
        return;
    }
    // a pure decode forward depends on device state only through these buffers: replayable (step_graph.h)
    decode_graph_.run({inp.data(), lengths.data(), new_item_indices.data(), decoder_result.data(), emb_table.data(),
                       pos_emb_table.data(), page_table.data()},
                      [&] { rounds(0); });
}

PagedAttentionCublasInferenceModel::PagedAttentionCublasInferenceModel(
    PagedAttentionCublasLayer&& attention_layer, PagedEncoderLayer&& encoder_layer,
    PagedCublasDecoderLayer&& decoder_layer, size_t n_batch, size_t n_sequence, size_t emb_dim, int n_forward_rounds)
    : paged_attention_layer_(std::move(attention_layer)), paged_encoder_layer_(std::move(encoder_layer)),
      paged_decoder_layer_(std::move(decoder_layer)), n_batch_(n_batch), n_sequence_(n_sequence),
      emb_dim_(emb_dim), attention_result_(std::vector<size_t>{n_batch, emb_dim}, DeviceType::DEVICE),
      n_forward_rounds_(n_forward_rounds) {}

void PagedAttentionCublasInferenceModel::forward(const TensorInt& inp, TensorInt& lengths,
                                                 const TensorInt& new_item_indices, TensorInt& decoder_result,
                                                 int n_new_items, const TensorFloat& emb_table,
                                                 const TensorFloat& pos_emb_table, TensorFloatPoint& page_table,
                                                 GemmHandle handle) {
    auto rounds = [&](int n_new) {
        for (int round = 0; round < n_forward_rounds_; ++round) {
            const int fresh = round == 0 ? n_new : 0;
            if (mli::runtime::lean_layers()) {
                paged_attention_layer_.prefill(emb_table, pos_emb_table, inp, page_table, lengths, new_item_indices, fresh);
                paged_attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, 0, handle);
            } else {
                paged_encoder_layer_.forward(emb_table, pos_emb_table, inp, page_table, lengths, new_item_indices, fresh);
                paged_attention_layer_.forward(page_table, lengths, new_item_indices, attention_result_, fresh, handle);
            }
            paged_decoder_layer_.forward(attention_result_, emb_table, pos_emb_table, page_table, lengths,
                                         decoder_result, round, handle);
        }
    };
    if (n_new_items != 0) {
        rounds(n_new_items);
        return;
    }
    decode_graph_.run({inp.data(), lengths.data(), new_item_indices.data(), decoder_result.data(), emb_table.data(),
                       pos_emb_table.data(), page_table.data()},
                      [&] { rounds(0); });
}
