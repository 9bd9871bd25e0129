// Engine loops (reference include/inferencer.h:18-32, src/inferencer.cpp): until every item is finished,
//   forward -> process_decoder_result -> (return / grow / preempt pages) -> insert_new_items.
// One engine drives one GPU (the calling thread's current device, runtime.h).
#pragma once

#include "inference_model.h"
#include "item_storage.h"
#include "paged_item_storage.h"
#include "tensor.hpp"

void start_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
                            ProcessingStorage& processing_storage, InferenceModel& inference_model,
                            size_t n_batch_size, size_t n_sequence);

void start_paged_attention_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                            ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                            MemoryBlockManager& memory_block_manager,
                                            PagedAttentionsManager& paged_attention_manager,
                                            PagedAttentionInferenceModel& inference_model, size_t n_batch_size,
                                            size_t n_sequence, int n_forward_rounds);

void start_paged_attention_cublas_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                   ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                   MemoryBlockManager& memory_block_manager,
                                                   PagedAttentionsManager& paged_attention_manager,
                                                   PagedAttentionCublasInferenceModel& inference_model,
                                                   size_t n_batch_size, size_t n_sequence, int n_forward_rounds);
