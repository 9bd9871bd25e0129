// KV page pool, per-row page lists and the host mirror of the device page table
// (reference include/paged_item_storage.h, src/paged_item_storage.cpp).
#pragma once

#include <cstddef>
#include <list>
#include <utility>
#include <vector>

#include "item_storage.h"
#include "tensor.hpp"

// One device allocation of n_blocks * each_block_size floats, handed out in fixed-size blocks.
// each_block_size = PAGE_BLOCK_SIZE * 3 * emb_dim (input embedding | K | V per token).
class MemoryBlockManager {
public:
    MemoryBlockManager(int n_blocks, size_t each_block_size);
    int free_blocks_size() const;
    std::list<float*> pop_free_blocks(int size);  // throws std::runtime_error when fewer are free
    void return_free_blocks(std::list<float*>&&);

private:
    TensorFloat block_memory_;
    std::list<float*> free_blocks_;
};

using BatchIdMemoryBlocksPair = std::pair<int, std::list<float*>>;

class PagedAttentionsManager {
public:
    PagedAttentionsManager(size_t max_batches, size_t n_sequence, size_t emb_dim);
    std::list<BatchIdMemoryBlocksPair>& get_used_block_list();
    // host page table -> device, only if something changed.  The reference re-uploads the whole table
    // (src/paged_item_storage.cpp:181-186: 2 MiB at B=1024, S=4096, every iteration in which any row crosses a page
    // boundary); here a few changed entries are scattered by one small kernel and only a bulk change (e.g. the
    // initial admission of a whole batch) copies the table.
    void maybe_flush_changes();
    void add_batch_block_pair(BatchIdMemoryBlocksPair&&);
    void set_block_pos(int batch_id, int i_block, float*);
    TensorFloatPoint& get_page_table_device();
    int max_blocks_per_row() const { return static_cast<int>(width_); }  // extension: page-table width
    // extension: the reference's stale-length upload (see set_reference_length_reset_quirk below), per manager so
    // that several engines in one process do not change each other's admission behaviour
    void set_length_reset_quirk(bool enabled) { length_reset_quirk_ = enabled; }
    bool length_reset_quirk() const { return length_reset_quirk_; }

private:
    TensorFloatPoint page_table_host;
    TensorFloatPoint page_table_device;
    std::list<BatchIdMemoryBlocksPair> used_blocks_;  // rows in admission order; the tail is preempted first
    size_t width_;                                    // n_sequence / PAGE_BLOCK_SIZE
    bool needs_sync_;
    bool length_reset_quirk_;
    std::vector<long long> dirty_;                    // flat indices of entries changed since the last flush
};

// Gives the row one more page (and records it in the host page table).
void allocate_memory_block(MemoryBlockManager&, PagedAttentionsManager&, BatchIdMemoryBlocksPair&);

// Returns the pages of finished rows, grows rows that are about to cross a page boundary, and preempts
// rows from the tail of the admission list (back to the head of the queue) when the pool is empty.
// last_row_short (extension, for the pipelined loop's look-ahead): when given, the ONLY row left in flight is not
// preempted when the pool cannot grow it -- *last_row_short is set instead and the row keeps its pages, so the caller
// can first learn whether the row needs the page at all (it may have finished in the forward that is still running).
void allocate_or_free_memory_blocks_if_needed(PagedAttentionsManager&, MemoryBlockManager&, ProcessingStorage&,
                                              ItemStorage&, const std::vector<int>& finished_indices,
                                              int n_forward_rounds, bool* last_row_short = nullptr);

// The host-only half of the paged insert: admission decisions, page hand-out, host mirrors (inp rows, lengths,
// new_idx[0 .. slots.size())) -- no device traffic.  insert_new_items = this + the uploads; the pipelined engine
// (pipelined_engine.h) pairs it with per-slot device updates instead.
struct PagedAdmission {
    std::vector<int> slots;        // slots filled, in slot order
    bool lengths_changed = false;  // some slot's length differs from what the device holds
};
PagedAdmission admit_new_items(int* inp_host, int* lengths_host, int* new_idx_host, int max_batch, int n_sequence,
                               ItemStorage& item_storage, ProcessingStorage& processing_storage,
                               MemoryBlockManager& memory_block_manager,
                               PagedAttentionsManager& paged_attention_manager, int n_forward_rounds);

// Paged engine: admit queued items into free slots while pages last; returns the slots filled.
std::vector<int> insert_new_items(TensorInt& inp_device, TensorInt& inp_host, TensorInt& lengths_device,
                                  TensorInt& lengths_host, TensorInt& new_items_indices_device,
                                  TensorInt& new_items_indices_host, ItemStorage& item_storage,
                                  ProcessingStorage& processing_storage, MemoryBlockManager& memory_block_manager,
                                  PagedAttentionsManager& paged_attention_manager, int n_forward_rounds);

// The reference's paged insert_new_items uploads lengths_host without ever refreshing it from the device
// (src/paged_item_storage.cpp:110-118), which resets every in-flight row to its insertion-time length
// whenever a slot is free.  Default here: in-flight rows keep their true length.  Set to true to
// reproduce the reference's behaviour (for measurement only).  The process-wide switch is only the value a
// PagedAttentionsManager built afterwards starts with; PagedAttentionsManager::set_length_reset_quirk changes one
// engine's behaviour without touching the others.
void set_reference_length_reset_quirk(bool enabled);
bool reference_length_reset_quirk();
