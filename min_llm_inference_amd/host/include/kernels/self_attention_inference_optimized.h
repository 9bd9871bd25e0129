// Contiguous-KV decode attention: the five launchers and their composition, with the reference's
// signatures (include/kernels/self_attention_inference_optimized.h:5-48).  Each is a thin adapter over
// the C ABI in include/mli_kernels.h.
//
// Shapes: inp_embedding [n_batch, n_sequence, input_dim]; lengths [n_batch]; wk/wq/wv [input_dim, output_dim];
// new_batch_idx [n_batch] (first n_new_items valid); kt_cache [n_batch, output_dim, n_sequence];
// v_cache [n_batch, n_sequence, output_dim]; q_output [n_batch, output_dim]; qkt_output [n_batch, n_sequence];
// attention_result [n_batch, output_dim].
#pragma once
// (extension at the end of this header: launch_prefill = encoder + launch_fill_new_kt_v_cache in one launch)

#include "tensor.hpp"

void launch_fill_new_kt_v_cache(const TensorFloat& inp_embedding, const TensorInt& new_batch_idx,
                                const TensorInt& lengths, const TensorFloat& wk, const TensorFloat& wv,
                                TensorFloat& kt_cache, TensorFloat& v_cache, int n_new_items);

void launch_get_latest_kt_q_v(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorFloat& wk,
                              const TensorFloat& wq, const TensorFloat& wv, TensorFloat& kt_cache,
                              TensorFloat& v_cache, TensorFloat& q_output);

void launch_qkt(const TensorFloat& q_output, const TensorFloat& kt_cache, const TensorInt& lengths,
                TensorFloat& qkt_output);

void launch_softmax_in_place_with_lengths(TensorFloat& qkt_output, const TensorInt& lengths);

void launch_softmax_v(const TensorFloat& softmax_result, const TensorFloat& v_cache, TensorFloat& attention_result,
                      const TensorInt& lengths);

void inference_self_attention(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorFloat& wk,
                              const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                              TensorFloat& kt_cache, TensorFloat& v_cache, TensorFloat& q_output,
                              TensorFloat& qkt_output, TensorFloat& attention_result, int n_new_items);

// EXTENSION: the composition without the scores (include/mli_kernels.h: mli_self_attention_lean); shapes its single-launch
// scan does not cover take inference_self_attention through the caller's qkt_output scratch.  What
// SelfAttentionLayer::forward runs unless runtime::set_lean_layers(false).
void inference_self_attention_lean(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorFloat& wk,
                                   const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                                   TensorFloat& kt_cache, TensorFloat& v_cache, TensorFloat& q_output,
                                   TensorFloat& qkt_output, TensorFloat& attention_result, int n_new_items);

// EXTENSION: launch_inference_optimized_encoder_kernel + launch_fill_new_kt_v_cache in one launch --
// the embedding lookup is the fill GEMM's prologue; inp_embedding, kt_cache and v_cache bit-identical to the two launches.
void launch_prefill(const TensorFloat& emb_table, const TensorFloat& wpe, const TensorInt& inp,
                    TensorFloat& inp_embedding, const TensorInt& lengths, const TensorInt& new_item_indices,
                    const TensorFloat& wk, const TensorFloat& wv, TensorFloat& kt_cache, TensorFloat& v_cache,
                    int n_new_items);
