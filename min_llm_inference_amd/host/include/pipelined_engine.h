// Pipelined paged engine loop.  EXTENSION -- SURVEY 8(f) row 3 ("device-resident batch insert"): the reference's loop
// (src/inferencer.cpp:43-122) is strictly sequential per iteration,
//     forward(k) -> D2H + process result(k) -> page bookkeeping -> insert + uploads -> forward(k+1),
// so the GPU idles while the host works.  Here the host runs ONE STEP BEHIND the GPU:
//     queue D2H(result k) | bookkeeping + admission for forward(k+1) | launch forward(k+1) | wait + process result(k)
// which is possible because
//   * the decoder kernel maintains the device lengths itself (L+1, or 0 when the row finishes), so a continuing row
//     needs no host input between steps;
//   * page growth depends only on the row's length, which the host knows one step ahead (tokens so far + the one in
//     flight);
//   * everything the host sends is a per-slot update that is stream-ordered and never blocks: new prompts by async
//     row copies from pinned memory, new lengths / preemptions / page-table entries by scatter kernels whose payload
//     travels in the kernel arguments (Tensor::scatter_from_host) -- the whole-tensor lengths upload of the
//     sequential loop would overwrite lengths the device has already advanced.
// Differences an observer can see: a freed slot is refilled one iteration later (its result is known one iteration
// later), and a row preempted while its token is in flight has that token dropped and regenerated after
// re-admission.  Per-item token streams are identical to the sequential engines' (greedy decoding is per row;
// tests/test_engine_gpu.py).  n_forward_rounds up to PAGE_BLOCK_SIZE / 2: a row then has up to R tokens in flight.
#pragma once

#include <functional>

#include "bf16_extension.h"
#include "fp8_extension.h"
#include "inferencer.h"

// (inp, lengths, new_item_indices, decoder_result, n_new_items): one forward of any paged model
using PagedForward = std::function<void(const TensorInt&, TensorInt&, const TensorInt&, TensorInt&, int)>;

// Returns the number of forward launches.
long long run_paged_engine_pipelined(ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                     MemoryBlockManager& memory_block_manager,
                                     PagedAttentionsManager& paged_attention_manager, size_t n_batch_size,
                                     size_t n_sequence, const PagedForward& forward, int n_forward_rounds = 1);

void start_paged_attention_inference_engine_pipelined(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                      ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                      MemoryBlockManager& memory_block_manager,
                                                      PagedAttentionsManager& paged_attention_manager,
                                                      PagedAttentionInferenceModel& inference_model,
                                                      size_t n_batch_size, size_t n_sequence, int n_forward_rounds = 1);

void start_paged_attention_cublas_inference_engine_pipelined(
    const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
    ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
    PagedAttentionsManager& paged_attention_manager, PagedAttentionCublasInferenceModel& inference_model,
    size_t n_batch_size, size_t n_sequence, int n_forward_rounds = 1);

void start_paged_attention_fp8_inference_engine_pipelined(
    const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
    ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
    PagedAttentionsManager& paged_attention_manager, PagedAttentionFp8InferenceModel& inference_model,
    size_t n_batch_size, size_t n_sequence, int n_forward_rounds = 1);

void start_paged_attention_bf16_inference_engine_pipelined(
    const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
    ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
    PagedAttentionsManager& paged_attention_manager, PagedAttentionBf16InferenceModel& inference_model,
    size_t n_batch_size, size_t n_sequence, int n_forward_rounds = 1);
