// Layer objects: own the weights and the pre-allocated scratch, unpack shapes, call the launchers.
// Same constructors and forward() signatures as the reference (include/layers.h:19-156) for the layers
// on the inference path.  FeedForward (used by no model in the reference) is out of scope.
#pragma once

#include "kernels/paged_attention.h"
#include "tensor.hpp"
#include "utils.h"

class SelfAttentionLayer : public NonCopyableNonClonable {
public:
    SelfAttentionLayer(TensorFloat&& wk, TensorFloat&& wq, TensorFloat&& wv, size_t n_batch, size_t input_dim,
                       size_t n_sequence);
    void forward(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorInt& new_batch_idx,
                 TensorFloat& attention_result, int n_new_items);
    // extension: encoder + K/V prefill of the new rows in one launch (the model then calls forward with n_new_items = 0)
    void prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloat& inp_embedding, const TensorInt& lengths, const TensorInt& new_item_indices, int n_new_items);

private:
    TensorFloat wk_, wq_, wv_;
    TensorFloat kt_cache_;    // [n_batch, input_dim, n_sequence]
    TensorFloat v_cache_;     // [n_batch, n_sequence, input_dim]
    TensorFloat q_output_;    // [n_batch, input_dim]
    TensorFloat qkt_output_;  // [n_batch, n_sequence]
};

class PagedAttentionLayer : public NonCopyableNonClonable {
public:
    PagedAttentionLayer(TensorFloat&& wk, TensorFloat&& wq, TensorFloat&& wv, size_t n_batch, size_t emb_dim,
                        size_t n_sequence);
    void forward(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_batch_idx,
                 TensorFloat& attention_result, int n_new_items);
    // extension: encoder + K/V prefill of the new rows in one launch (the model then calls forward with n_new_items = 0)
    void prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                 int n_new_items);

private:
    TensorFloat wk_, wq_, wv_;
    TensorFloat q_output_;
    TensorFloat qkt_output_;
};

class PagedAttentionCublasLayer : public NonCopyableNonClonable {
public:
    PagedAttentionCublasLayer(TensorFloat&& wk, TensorFloat&& wq, TensorFloat&& wv, size_t n_batch, size_t emb_dim,
                              size_t n_sequence);
    void forward(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_batch_idx,
                 TensorFloat& attention_result, int n_new_items, GemmHandle& handle);
    void prefill(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                 int n_new_items);  // extension, as PagedAttentionLayer::prefill

private:
    TensorFloat wk_, wq_, wv_;
    TensorFloat q_output_;
    TensorFloat qkt_output_;
    TensorFloat latest_emb_;        // kept for signature parity with the reference; unused by the MFMA path
    TensorFloat temp_placeholder_;
};

// emb_table [n_vocab, dim], pos_emb [n_sequence, dim], inp [n_batch, n_sequence] token ids
class EncoderLayer : public NonCopyableNonClonable {
public:
    void forward(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloat& inp_embedding, const TensorInt& lengths, const TensorInt& new_item_indices,
                 int n_new_items);
};

class PagedEncoderLayer : public NonCopyableNonClonable {
public:
    void forward(const TensorFloat& emb_table, const TensorFloat& pos_emb, const TensorInt& inp,
                 TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                 int n_new_items);
};

class DecoderLayer : public NonCopyableNonClonable {
public:
    DecoderLayer(size_t n_batch, size_t n_vocab);
    void forward(const TensorFloat& batch_result, const TensorFloat& emb_table, const TensorFloat& wpe_table,
                 TensorFloat& inp_embedding, TensorInt& lengths, TensorInt& decoder_result);

private:
    TensorFloat emb_score_;  // [n_batch, n_vocab]
};

class PagedDecoderLayer : public NonCopyableNonClonable {
public:
    PagedDecoderLayer(size_t n_batch, size_t n_vocab);
    void forward(const TensorFloat& batch_result, const TensorFloat& emb_table, const TensorFloat& wpe_table,
                 TensorFloatPoint& page_table, TensorInt& lengths, TensorInt& decoder_result, int i_decoder_round);

private:
    TensorFloat emb_score_;
};

class PagedCublasDecoderLayer : public NonCopyableNonClonable {
public:
    PagedCublasDecoderLayer(size_t n_batch, size_t n_vocab);
    void forward(const TensorFloat& batch_result, const TensorFloat& emb_table, const TensorFloat& wpe_table,
                 TensorFloatPoint& page_table, TensorInt& lengths, TensorInt& decoder_result, int i_decoder_round,
                 GemmHandle& handle);

private:
    TensorFloat emb_score_;
};
