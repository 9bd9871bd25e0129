// Greedy decoder head (reference include/kernels/decoder.h:19-37): logits = batch_result . emb_table^T,
// per-row argmax, lengths update (0 when EOF or the row is full), next token's embedding written at
// position lengths[b].
#pragma once

#include "kernels/paged_attention.h"
#include "tensor.hpp"

void launch_decoder(const TensorFloat& batch_result, const TensorFloat& emb_table, TensorFloat& emb_score,
                    const TensorFloat& wpe_table, TensorFloat& inp_embedding, TensorInt& lengths,
                    TensorInt& decoder_result);

void launch_paged_attention_decoder_multi_rounds(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                                 TensorFloat& emb_score, const TensorFloat& wpe_table,
                                                 TensorFloatPoint& page_table, TensorInt& lengths,
                                                 TensorInt& decoder_result, int i_decoder);

// EXTENSION (no reference counterpart): the same decoder head with the argmax as the logits GEMM's epilogue: same
// tokens, lengths and embeddings; emb_score is lent as scratch and holds no scores afterwards.  What the decoder
// layers run unless runtime::set_lean_layers(false).
void launch_decoder_fused(const TensorFloat& batch_result, const TensorFloat& emb_table, TensorFloat& emb_score,
                          const TensorFloat& wpe_table, TensorFloat& inp_embedding, TensorInt& lengths,
                          TensorInt& decoder_result);
void launch_paged_attention_decoder_fused(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                          TensorFloat& emb_score, const TensorFloat& wpe_table,
                                          TensorFloatPoint& page_table, TensorInt& lengths, TensorInt& decoder_result,
                                          int i_decoder);

void launch_paged_attention_cublas_decoder_multi_rounds(const TensorFloat& batch_result,
                                                        const TensorFloat& emb_table, TensorFloat& emb_score,
                                                        const TensorFloat& wpe_table, TensorFloatPoint& page_table,
                                                        TensorInt& lengths, TensorInt& decoder_result, int i_decoder,
                                                        GemmHandle& handle);
