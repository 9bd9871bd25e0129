#include "inferencer.h"

#include <cstring>
#include <numeric>
#include <stdexcept>
#include <vector>

#include "bf16_extension.h"
#include "constants.h"
#include "fp8_extension.h"
#include "pipelined_engine.h"
#include "runtime.h"
#include "throughput_counter.h"

namespace {

// Host/device pairs every engine loop needs; HOST tensors are pinned.
struct LoopTensors {
    TensorInt inp_device, inp_host;
    TensorInt lengths_device, lengths_host;
    TensorInt new_items_indices_device, new_items_indices_host;
    TensorInt decoder_result_device, decoder_result_host;

    LoopTensors(size_t n_batch, size_t n_sequence, std::vector<size_t> result_shape)
        : inp_device({n_batch, n_sequence}, DeviceType::DEVICE), inp_host({n_batch, n_sequence}, DeviceType::HOST),
          lengths_device({n_batch}, DeviceType::DEVICE), lengths_host({n_batch}, DeviceType::HOST),
          new_items_indices_device({n_batch}, DeviceType::DEVICE), new_items_indices_host({n_batch}, DeviceType::HOST),
          decoder_result_device(result_shape, DeviceType::DEVICE), decoder_result_host(result_shape, DeviceType::HOST) {}
};

struct Range {  // roctx range, closed on scope exit
    explicit Range(const char* name) { mli::runtime::range_push(name); }
    ~Range() { mli::runtime::range_pop(); }
};

// Nothing in flight, items still queued, and the admission just refused the head item although every page is free:
// the pool cannot hold it.  The reference's loop spins forever here; a host waiting on a GPU box should get an error.
void throw_if_stuck(ItemStorage& item_storage, ProcessingStorage& processing_storage) {
    if (processing_storage.size() == 0 && item_storage.new_count() > 0)
        throw std::runtime_error("paged engine: the page pool is too small for the next queued item");
}

// Shared body of the two paged engines; `forward` hides the model type (and its GemmHandle).
template <typename Forward>
void run_paged_engine(ItemStorage& item_storage, ProcessingStorage& processing_storage,
                      MemoryBlockManager& memory_block_manager, PagedAttentionsManager& paged_attention_manager,
                      size_t n_batch_size, size_t n_sequence, int n_forward_rounds, Forward&& forward) {
    LoopTensors t(n_batch_size, n_sequence, {n_batch_size, static_cast<size_t>(n_forward_rounds)});
    // Every slot starts empty on both sides.  The reference's first insert writes and uploads every slot; here an
    // insert uploads only what changed, so the mirrors (pinned memory) and the device tensors (hipMalloc) must not
    // start with whatever the allocations held: a forward over garbage lengths would chase garbage page pointers.
    std::memset(t.lengths_host.data(), 0, n_batch_size * sizeof(int));
    std::memset(t.inp_host.data(), 0, n_batch_size * n_sequence * sizeof(int));
    t.lengths_device.copy_from(t.lengths_host);
    t.inp_device.copy_from(t.inp_host);
    get_global_throughput_counter().start_record();
    std::vector<int> new_item_indices;
    {
        Range r("insert_new_items");
        new_item_indices = insert_new_items(t.inp_device, t.inp_host, t.lengths_device, t.lengths_host,
                                            t.new_items_indices_device, t.new_items_indices_host, item_storage,
                                            processing_storage, memory_block_manager, paged_attention_manager,
                                            n_forward_rounds);
    }
    throw_if_stuck(item_storage, processing_storage);  // no forward is launched for a queue nothing can be admitted from
    while (!is_done(item_storage, processing_storage)) {
        {
            Range r("forward");
            forward(t, static_cast<int>(new_item_indices.size()));
        }
        std::vector<int> finished_indices;
        {
            Range r("process_decoder_result");
            finished_indices = process_decoder_result(t.decoder_result_device, t.decoder_result_host, item_storage,
                                                      processing_storage, static_cast<int>(n_sequence));
        }
        {
            Range r("allocate_or_free_memory_blocks_if_needed");
            allocate_or_free_memory_blocks_if_needed(paged_attention_manager, memory_block_manager,
                                                     processing_storage, item_storage, finished_indices,
                                                     n_forward_rounds);
        }
        {
            Range r("insert_new_items");
            new_item_indices = insert_new_items(t.inp_device, t.inp_host, t.lengths_device, t.lengths_host,
                                                t.new_items_indices_device, t.new_items_indices_host, item_storage,
                                                processing_storage, memory_block_manager, paged_attention_manager,
                                                n_forward_rounds);
        }
        throw_if_stuck(item_storage, processing_storage);
    }
    get_global_throughput_counter().print_throughput();
}


// the pipelined loop is the default wherever it applies (runtime.h: set_sequential_engine_loop)
bool pipelined_loop_applies(const PagedAttentionsManager& pages, int n_forward_rounds) {
    return !mli::runtime::sequential_engine_loop() && 2 * n_forward_rounds <= PAGE_BLOCK_SIZE && !pages.length_reset_quirk();
}

}  // namespace

void start_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table, ItemStorage& item_storage,
                            ProcessingStorage& processing_storage, InferenceModel& inference_model,
                            size_t n_batch_size, size_t n_sequence) {
    LoopTensors t(n_batch_size, n_sequence, {n_batch_size});
    std::vector<int> free_slots(n_batch_size);
    std::iota(free_slots.begin(), free_slots.end(), 0);  // every slot starts empty
    int n_new_items = insert_new_items(free_slots, t.inp_device, t.inp_host, t.lengths_device, t.lengths_host,
                                       t.new_items_indices_device, t.new_items_indices_host, item_storage,
                                       processing_storage);
    while (!is_done(item_storage, processing_storage)) {
        inference_model.forward(t.inp_device, t.lengths_device, t.new_items_indices_device, t.decoder_result_device,
                                n_new_items, emb_table, pos_table);
        free_slots = process_decoder_result(t.decoder_result_device, t.decoder_result_host, item_storage,
                                            processing_storage, static_cast<int>(n_sequence));
        n_new_items = insert_new_items(free_slots, t.inp_device, t.inp_host, t.lengths_device, t.lengths_host,
                                       t.new_items_indices_device, t.new_items_indices_host, item_storage,
                                       processing_storage);
    }
}

void start_paged_attention_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                            ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                            MemoryBlockManager& memory_block_manager,
                                            PagedAttentionsManager& paged_attention_manager,
                                            PagedAttentionInferenceModel& inference_model, size_t n_batch_size,
                                            size_t n_sequence, int n_forward_rounds) {
    if (pipelined_loop_applies(paged_attention_manager, n_forward_rounds)) {
        start_paged_attention_inference_engine_pipelined(emb_table, pos_table, item_storage, processing_storage,
                                                         memory_block_manager, paged_attention_manager, inference_model,
                                                         n_batch_size, n_sequence, n_forward_rounds);
        return;
    }
    run_paged_engine(item_storage, processing_storage, memory_block_manager, paged_attention_manager, n_batch_size,
                     n_sequence, n_forward_rounds, [&](LoopTensors& t, int n_new_items) {
                         inference_model.forward(t.inp_device, t.lengths_device, t.new_items_indices_device,
                                                 t.decoder_result_device, n_new_items, emb_table, pos_table,
                                                 paged_attention_manager.get_page_table_device());
                     });
}

void start_paged_attention_cublas_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                   ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                   MemoryBlockManager& memory_block_manager,
                                                   PagedAttentionsManager& paged_attention_manager,
                                                   PagedAttentionCublasInferenceModel& inference_model,
                                                   size_t n_batch_size, size_t n_sequence, int n_forward_rounds) {
    if (pipelined_loop_applies(paged_attention_manager, n_forward_rounds)) {
        start_paged_attention_cublas_inference_engine_pipelined(emb_table, pos_table, item_storage, processing_storage,
                                                                memory_block_manager, paged_attention_manager,
                                                                inference_model, n_batch_size, n_sequence, n_forward_rounds);
        return;
    }
    GemmHandle handle;  // the reference creates / destroys a cublasHandle_t here; nothing to create for MFMA
    run_paged_engine(item_storage, processing_storage, memory_block_manager, paged_attention_manager, n_batch_size,
                     n_sequence, n_forward_rounds, [&](LoopTensors& t, int n_new_items) {
                         inference_model.forward(t.inp_device, t.lengths_device, t.new_items_indices_device,
                                                 t.decoder_result_device, n_new_items, emb_table, pos_table,
                                                 paged_attention_manager.get_page_table_device(), handle);
                     });
}

// EXTENSION (no reference counterpart): the same loop over bf16 pages and weights.
void start_paged_attention_bf16_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                 ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                 MemoryBlockManager& memory_block_manager,
                                                 PagedAttentionsManager& paged_attention_manager,
                                                 PagedAttentionBf16InferenceModel& inference_model,
                                                 size_t n_batch_size, size_t n_sequence, int n_forward_rounds) {
    if (pipelined_loop_applies(paged_attention_manager, n_forward_rounds)) {
        start_paged_attention_bf16_inference_engine_pipelined(emb_table, pos_table, item_storage, processing_storage,
                                                              memory_block_manager, paged_attention_manager,
                                                              inference_model, n_batch_size, n_sequence, n_forward_rounds);
        return;
    }
    run_paged_engine(item_storage, processing_storage, memory_block_manager, paged_attention_manager, n_batch_size,
                     n_sequence, n_forward_rounds, [&](LoopTensors& t, int n_new_items) {
                         inference_model.forward(t.inp_device, t.lengths_device, t.new_items_indices_device,
                                                 t.decoder_result_device, n_new_items, emb_table, pos_table,
                                                 paged_attention_manager.get_page_table_device());
                     });
}

// EXTENSION, opt-in (no reference counterpart): the same loop over fp8 (OCP e4m3) pages and bf16 weights.
void start_paged_attention_fp8_inference_engine(const TensorFloat& emb_table, const TensorFloat& pos_table,
                                                ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                                MemoryBlockManager& memory_block_manager,
                                                PagedAttentionsManager& paged_attention_manager,
                                                PagedAttentionFp8InferenceModel& inference_model, size_t n_batch_size,
                                                size_t n_sequence, int n_forward_rounds) {
    if (pipelined_loop_applies(paged_attention_manager, n_forward_rounds)) {
        start_paged_attention_fp8_inference_engine_pipelined(emb_table, pos_table, item_storage, processing_storage,
                                                             memory_block_manager, paged_attention_manager,
                                                             inference_model, n_batch_size, n_sequence, n_forward_rounds);
        return;
    }
    run_paged_engine(item_storage, processing_storage, memory_block_manager, paged_attention_manager, n_batch_size,
                     n_sequence, n_forward_rounds, [&](LoopTensors& t, int n_new_items) {
                         inference_model.forward(t.inp_device, t.lengths_device, t.new_items_indices_device,
                                                 t.decoder_result_device, n_new_items, emb_table, pos_table,
                                                 paged_attention_manager.get_page_table_device());
                     });
}
