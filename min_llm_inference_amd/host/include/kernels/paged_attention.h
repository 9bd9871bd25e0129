// Paged-KV decode attention launchers with the reference's signatures
// (include/kernels/paged_attention.h:17-67).  page_table is [n_batch, n_sequence / PAGE_BLOCK_SIZE] device
// pointers to blocks of PAGE_BLOCK_SIZE * 3 * emb_dim floats (input embedding | K | V per token).
#pragma once

#include "tensor.hpp"

// The reference threads a cublasHandle_t through its cuBLAS variants.  The MFMA kernels need no library
// handle; GemmHandle is an empty tag that keeps the parameter position, so a call site changes one type name.
struct GemmHandle {};

void paged_attention(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorFloat& wk,
                     const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                     TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result, int n_new_items,
                     int n_sequence);

void launch_fill_new_k_v_cache_paged_attention(TensorFloatPoint page_table, const TensorInt& new_batch_idx,
                                               const TensorInt& lengths, const TensorFloat& wk,
                                               const TensorFloat& wv, int n_new_items, int n_sequence);

void launch_get_latest_k_q_v_paged_attention(TensorFloatPoint& page_table, const TensorInt& lengths,
                                             const TensorFloat& wk, const TensorFloat& wq, const TensorFloat& wv,
                                             TensorFloat& q_output, int n_sequence);

void launch_qkt_paged_attention(const TensorFloat& q_output, const TensorFloatPoint& page_table,
                                const TensorInt& lengths, TensorFloat& qkt_output);

void launch_softmax_v_paged_attention(const TensorFloat& softmax_result, const TensorFloatPoint& page_table,
                                      TensorFloat& attention_result, const TensorInt& lengths);

// EXTENSION (no reference counterpart): paged_attention without materialising scores / probabilities -- q_output and
// attention_result come out bit-identical, qkt_output is left alone (it is only used when emb_dim exceeds what the
// single-pass kernel covers).  What PagedAttention[Cublas]Layer::forward runs unless runtime::set_lean_layers(false).
void paged_attention_lean(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorFloat& wk,
                          const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                          TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result,
                          int n_new_items, int n_sequence);

// EXTENSION: launch_paged_attention_encoder_kernel + launch_fill_new_k_v_cache_paged_attention in one
// launch -- the embedding lookup is the fill GEMM's prologue; pages bit-identical to the two-launch form.
void launch_paged_prefill(const TensorFloat& emb_table, const TensorFloat& wpe, const TensorInt& inp,
                          TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                          const TensorFloat& wk, const TensorFloat& wv, int n_new_items);

// "cuBLAS" variants: same results, produced by the same gather-GEMM-scatter MFMA kernel.  latest_emb and
// temp_placeholder were scratch for the three cublasSgemm calls and are accepted but not used.
void paged_attention_with_cublas(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorFloat& wk,
                                 const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                                 TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result,
                                 TensorFloat& latest_emb, TensorFloat& temp_placeholder, int n_new_items,
                                 int n_sequence, GemmHandle& handle);

void launch_get_latest_k_q_v_paged_attention_cublas(TensorFloatPoint& page_table, const TensorInt& lengths,
                                                    TensorFloat& latest_emb, const TensorFloat& wk,
                                                    const TensorFloat& wq, const TensorFloat& wv,
                                                    TensorFloat& q_output, TensorFloat& temp_placeholder,
                                                    GemmHandle& handle, int n_sequence);

void launch_fill_new_k_v_cache_paged_attention_warp_tiling(TensorFloatPoint page_table,
                                                           const TensorInt& new_batch_idx, const TensorInt& lengths,
                                                           const TensorFloat& wk, const TensorFloat& wv,
                                                           int n_new_items, int n_sequence);
