#include "paged_item_storage.h"

#include <algorithm>
#include <cassert>
#include <iterator>
#include <stdexcept>

#include "constants.h"
#include "utils.h"

namespace {
bool g_length_reset_quirk = false;  // only the DEFAULT a newly built PagedAttentionsManager starts with
}

void set_reference_length_reset_quirk(bool enabled) { g_length_reset_quirk = enabled; }
bool reference_length_reset_quirk() { return g_length_reset_quirk; }

// ---- MemoryBlockManager -------------------------------------------------------------------------------
MemoryBlockManager::MemoryBlockManager(int n_blocks, size_t each_block_size)
    : block_memory_(std::vector<size_t>{static_cast<size_t>(n_blocks) * each_block_size}, DeviceType::DEVICE) {
    float* base = block_memory_.data();
    for (int i = 0; i < n_blocks; ++i) free_blocks_.push_back(base + static_cast<size_t>(i) * each_block_size);
}

int MemoryBlockManager::free_blocks_size() const { return static_cast<int>(free_blocks_.size()); }

std::list<float*> MemoryBlockManager::pop_free_blocks(int size) {
    if (free_blocks_size() < size) throw std::runtime_error("No enough block memories to return");
    std::list<float*> taken;
    auto last = std::next(free_blocks_.begin(), size);
    taken.splice(taken.end(), free_blocks_, free_blocks_.begin(), last);
    return taken;
}

void MemoryBlockManager::return_free_blocks(std::list<float*>&& blocks) {
    free_blocks_.splice(free_blocks_.end(), blocks);
}

// ---- PagedAttentionsManager ---------------------------------------------------------------------------
PagedAttentionsManager::PagedAttentionsManager(size_t max_batches, size_t n_sequence, size_t /*emb_dim*/)
    : page_table_host(std::vector<size_t>{max_batches, n_sequence / PAGE_BLOCK_SIZE}, DeviceType::HOST),
      page_table_device(std::vector<size_t>{max_batches, n_sequence / PAGE_BLOCK_SIZE}, DeviceType::DEVICE),
      width_(n_sequence / PAGE_BLOCK_SIZE), needs_sync_(false), length_reset_quirk_(g_length_reset_quirk) {
    assert(n_sequence % PAGE_BLOCK_SIZE == 0);
    // every entry starts null on both sides: a kernel that meets a row without pages skips it instead of
    // dereferencing whatever the allocation held
    std::fill(page_table_host.data(), page_table_host.data() + max_batches * width_, static_cast<float*>(nullptr));
    page_table_device.copy_from(page_table_host);
}

std::list<BatchIdMemoryBlocksPair>& PagedAttentionsManager::get_used_block_list() { return used_blocks_; }
TensorFloatPoint& PagedAttentionsManager::get_page_table_device() { return page_table_device; }

void PagedAttentionsManager::maybe_flush_changes() {
    constexpr size_t kScatterLimit = 512;  // beyond this one bulk copy is cheaper than the scatter launches
    if (needs_sync_) {
        if (dirty_.size() > kScatterLimit) {
            page_table_device.copy_from(page_table_host);
        } else {
            std::vector<float*> values(dirty_.size());
            for (size_t i = 0; i < dirty_.size(); ++i) values[i] = page_table_host.data()[dirty_[i]];
            page_table_device.scatter_from_host(dirty_.data(), values.data(), dirty_.size());
        }
    }
    dirty_.clear();
    needs_sync_ = false;
}

void PagedAttentionsManager::set_block_pos(int batch_id, int i_block, float* block) {
    const size_t at = static_cast<size_t>(batch_id) * width_ + i_block;
    page_table_host.data()[at] = block;
    dirty_.push_back(static_cast<long long>(at));
    needs_sync_ = true;
}

void PagedAttentionsManager::add_batch_block_pair(BatchIdMemoryBlocksPair&& row) {
    float** entry = page_table_host.data() + static_cast<size_t>(row.first) * width_;
    for (float* block : row.second) {
        dirty_.push_back(static_cast<long long>(entry - page_table_host.data()));
        *entry++ = block;
    }
    used_blocks_.push_back(std::move(row));
    needs_sync_ = true;
}

void allocate_memory_block(MemoryBlockManager& pool, PagedAttentionsManager& pages, BatchIdMemoryBlocksPair& row) {
    float* block = pool.pop_free_blocks(1).front();
    row.second.push_front(block);
    pages.set_block_pos(row.first, static_cast<int>(row.second.size()) - 1, block);
}

void allocate_or_free_memory_blocks_if_needed(PagedAttentionsManager& pages, MemoryBlockManager& pool,
                                              ProcessingStorage& processing_storage, ItemStorage& item_storage,
                                              const std::vector<int>& finished_indices, int n_forward_rounds,
                                              bool* last_row_short) {
    if (last_row_short != nullptr) *last_row_short = false;
    // a row needs at most one more page per iteration
    assert(n_forward_rounds > 0 && n_forward_rounds <= PAGE_BLOCK_SIZE);
    std::list<BatchIdMemoryBlocksPair>& rows = pages.get_used_block_list();

    // 1. finished rows hand their pages back
    // (finished_indices also lists slots that were already empty; they own no row)
    std::vector<int> sorted_copy;
    const std::vector<int>* done = &finished_indices;  // process_decoder_result returns slots in ascending order
    if (!std::is_sorted(done->begin(), done->end())) {
        sorted_copy = finished_indices;
        std::sort(sorted_copy.begin(), sorted_copy.end());
        done = &sorted_copy;
    }
    for (auto it = rows.begin(); !done->empty() && it != rows.end();) {
        if (std::binary_search(done->begin(), done->end(), it->first)) {
            pool.return_free_blocks(std::move(it->second));
            it = rows.erase(it);
        } else {
            ++it;
        }
    }

    // 2. rows whose next n_forward_rounds tokens no longer fit get one more page; when the pool is dry
    //    the most recently admitted row (list tail) is pushed back to the head of the queue
    for (auto it = rows.begin(); it != rows.end();) {
        assert(processing_storage.batch_id_processing(it->first));
        const size_t n_tokens = processing_storage.get_token(it->first).second.size();
        // A row never holds more than n_sequence tokens.  The reference does not cap this, so with
        // n_forward_rounds > 1 a row close to n_sequence asks for page index == table width and
        // set_block_pos() overwrites the next row's first entry (src/paged_item_storage.cpp:40,196-203);
        // its tests only ever run n_forward_rounds == 1.
        const size_t row_capacity = static_cast<size_t>(pages.max_blocks_per_row()) * PAGE_BLOCK_SIZE;
        const size_t needed = std::min(n_tokens + n_forward_rounds, row_capacity);
        if (needed <= it->second.size() * PAGE_BLOCK_SIZE) {
            ++it;
            continue;
        }
        if (pool.free_blocks_size() > 0) {
            allocate_memory_block(pool, pages, *it);  // re-checked on the next pass of the loop
        } else if (std::next(it) == rows.end()) {
            if (last_row_short != nullptr && it == rows.begin()) {  // the only row left: the caller decides
                *last_row_short = true;
                break;
            }
            processing_storage.move_to_new(it->first, item_storage);
            pool.return_free_blocks(std::move(it->second));
            it = rows.erase(it);
        } else {
            BatchIdMemoryBlocksPair victim(std::move(rows.back()));
            rows.pop_back();
            processing_storage.move_to_new(victim.first, item_storage);
            pool.return_free_blocks(std::move(victim.second));
        }
    }
}

PagedAdmission admit_new_items(int* inp, int* lengths, int* new_idx, int max_batch, int n_sequence,
                               ItemStorage& item_storage, ProcessingStorage& processing_storage,
                               MemoryBlockManager& pool, PagedAttentionsManager& pages, int n_forward_rounds) {
    assert(n_forward_rounds > 0 && n_forward_rounds <= PAGE_BLOCK_SIZE);
    std::vector<char> occupied(static_cast<size_t>(max_batch), 0);
    for (const BatchIdMemoryBlocksPair& row : pages.get_used_block_list()) occupied[row.first] = 1;

    const bool quirk = pages.length_reset_quirk();
    PagedAdmission result;
    for (int slot = 0; slot < max_batch; ++slot) {
        if (occupied[slot]) {
            // in-flight row: its device length equals its host token count (see src/item_storage.cpp);
            // the reference leaves the stale insertion-time value here (quirk, off by default)
            if (!quirk)
                lengths[slot] = static_cast<int>(processing_storage.get_token(slot).second.size());
            continue;
        }
        // A free slot whose mirrored length is already 0 was free at the last upload too, and the device agrees
        // (the decoder zeroes finished rows itself): nothing to upload for it.  A non-zero mirror means the row
        // left since then -- finished, or preempted with its device length still live -- so the lengths go up.
        // (The reference uploads whenever any slot is free; kept under the quirk switch.)
        if (lengths[slot] != 0 || quirk) result.lengths_changed = true;
        const int width = pages.max_blocks_per_row();
        const bool can_admit = pool.free_blocks_size() >= DEFAULT_INIT_NUM_BLOCKS && item_storage.new_count() > 0 &&
                               pool.free_blocks_size() >= std::min(width, ceil_div(item_storage.head_length() + n_forward_rounds, PAGE_BLOCK_SIZE));
        if (!can_admit) {
            lengths[slot] = 0;
            continue;
        }
        IdTokensPair item = std::move(item_storage.pop_new_items(1)[0]);
        const int n_tokens = static_cast<int>(item.second.size());
        assert(n_tokens + 1 <= n_sequence);
        lengths[slot] = n_tokens;
        result.lengths_changed = true;
        std::copy(item.second.begin(), item.second.end(), inp + static_cast<size_t>(slot) * n_sequence);
        new_idx[result.slots.size()] = slot;
        const int n_pages = std::min(width, std::max(ceil_div(n_tokens + n_forward_rounds, PAGE_BLOCK_SIZE), DEFAULT_INIT_NUM_BLOCKS));
        processing_storage.put(slot, std::move(item));
        pages.add_batch_block_pair(std::make_pair(slot, pool.pop_free_blocks(n_pages)));
        result.slots.push_back(slot);
    }
    return result;
}

std::vector<int> insert_new_items(TensorInt& inp_device, TensorInt& inp_host, TensorInt& lengths_device,
                                  TensorInt& lengths_host, TensorInt& new_items_indices_device,
                                  TensorInt& new_items_indices_host, ItemStorage& item_storage,
                                  ProcessingStorage& processing_storage, MemoryBlockManager& pool,
                                  PagedAttentionsManager& pages, int n_forward_rounds) {
    const int max_batch = static_cast<int>(inp_device.shape()[0]);
    const int n_sequence = static_cast<int>(inp_device.shape()[1]);
    PagedAdmission adm = admit_new_items(inp_host.data(), lengths_host.data(), new_items_indices_host.data(), max_batch,
                                         n_sequence, item_storage, processing_storage, pool, pages, n_forward_rounds);
    upload_changed_rows(inp_device, inp_host, adm.slots, lengths_host.data(), n_sequence);
    if (adm.lengths_changed) lengths_device.copy_from(lengths_host);
    if (!adm.slots.empty() || pages.length_reset_quirk()) new_items_indices_device.copy_from(new_items_indices_host);
    pages.maybe_flush_changes();
    return adm.slots;
}
