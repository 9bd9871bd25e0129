// The pipelined loop (pipelined_engine.cpp) bound to the three paged models.  Kept apart from the loop itself so that
// the loop links into the CPU scheduler tests without the model classes.
#include "pipelined_engine.h"

#include "throughput_counter.h"

void start_paged_attention_inference_engine_pipelined(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                      ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                      MemoryBlockManager& memory_block_manager,
                                                      PagedAttentionsManager& paged_attention_manager,
                                                      PagedAttentionInferenceModel& inference_model,
                                                      size_t n_batch_size, size_t n_sequence, int n_forward_rounds) {
    run_paged_engine_pipelined(item_storage, processing_storage, memory_block_manager, paged_attention_manager,
                               n_batch_size, n_sequence,
                               [&](const TensorInt& inp, TensorInt& lengths, const TensorInt& new_idx, TensorInt& result, int n_new) {
                                   inference_model.forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                                           paged_attention_manager.get_page_table_device());
                               }, n_forward_rounds);
    get_global_throughput_counter().print_throughput();
}

void start_paged_attention_cublas_inference_engine_pipelined(
    const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
    ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
    PagedAttentionsManager& paged_attention_manager, PagedAttentionCublasInferenceModel& inference_model,
    size_t n_batch_size, size_t n_sequence, int n_forward_rounds) {
    GemmHandle handle;
    run_paged_engine_pipelined(item_storage, processing_storage, memory_block_manager, paged_attention_manager,
                               n_batch_size, n_sequence,
                               [&](const TensorInt& inp, TensorInt& lengths, const TensorInt& new_idx, TensorInt& result, int n_new) {
                                   inference_model.forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                                           paged_attention_manager.get_page_table_device(), handle);
                               }, n_forward_rounds);
    get_global_throughput_counter().print_throughput();
}

void start_paged_attention_bf16_inference_engine_pipelined(
    const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
    ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
    PagedAttentionsManager& paged_attention_manager, PagedAttentionBf16InferenceModel& inference_model,
    size_t n_batch_size, size_t n_sequence, int n_forward_rounds) {
    run_paged_engine_pipelined(item_storage, processing_storage, memory_block_manager, paged_attention_manager,
                               n_batch_size, n_sequence,
                               [&](const TensorInt& inp, TensorInt& lengths, const TensorInt& new_idx, TensorInt& result, int n_new) {
                                   inference_model.forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                                           paged_attention_manager.get_page_table_device());
                               }, n_forward_rounds);
    get_global_throughput_counter().print_throughput();
}

void start_paged_attention_fp8_inference_engine_pipelined(
    const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
    ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
    PagedAttentionsManager& paged_attention_manager, PagedAttentionFp8InferenceModel& inference_model,
    size_t n_batch_size, size_t n_sequence, int n_forward_rounds) {
    run_paged_engine_pipelined(item_storage, processing_storage, memory_block_manager, paged_attention_manager,
                               n_batch_size, n_sequence,
                               [&](const TensorInt& inp, TensorInt& lengths, const TensorInt& new_idx, TensorInt& result, int n_new) {
                                   inference_model.forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                                           paged_attention_manager.get_page_table_device());
                               }, n_forward_rounds);
    get_global_throughput_counter().print_throughput();
}
