#include "item_storage.h"

#include <algorithm>
#include <cassert>
#include <iterator>

#include "constants.h"
#include "throughput_counter.h"

// ---- Storage ----------------------------------------------------------------------------------------
int Storage::size() const { return static_cast<int>(data_.size()); }
int Storage::head_length() const { return static_cast<int>(data_.front().second.size()); }
const IdTokensPair& Storage::get_top() const { return data_.front(); }
const std::list<IdTokensPair>& Storage::get_data() const { return data_; }
void Storage::add(IdTokensPair&& item) { data_.emplace_back(std::move(item)); }
void Storage::add_to_front(IdTokensPair&& item) { data_.emplace_front(std::move(item)); }

std::vector<IdTokensPair> Storage::pop_pairs(int size) {
    const int n = std::min(size, this->size());
    std::vector<IdTokensPair> out;
    out.reserve(n);
    for (int i = 0; i < n; ++i) {
        out.emplace_back(std::move(data_.front()));
        data_.pop_front();
    }
    return out;
}

// ---- ItemStorage ------------------------------------------------------------------------------------
std::vector<IdTokensPair> ItemStorage::pop_finished_items(int size) { return finished_items_.pop_pairs(size); }
std::vector<IdTokensPair> ItemStorage::pop_new_items(int size) { return new_items_.pop_pairs(size); }
const IdTokensPair& ItemStorage::get_top() const { return new_items_.get_top(); }
void ItemStorage::add_finished_item(IdTokensPair&& item) { finished_items_.add(std::move(item)); }
void ItemStorage::add_new_item(IdTokensPair&& item) { new_items_.add(std::move(item)); }
void ItemStorage::add_new_item_to_head(IdTokensPair&& item) { new_items_.add_to_front(std::move(item)); }
int ItemStorage::finish_count() const { return finished_items_.size(); }
int ItemStorage::new_count() const { return new_items_.size(); }
int ItemStorage::head_length() const { return new_items_.head_length(); }
const std::list<IdTokensPair>& ItemStorage::get_finished_items() const { return finished_items_.get_data(); }

// ---- ProcessingStorage ------------------------------------------------------------------------------
void ProcessingStorage::put(int batch_id, IdTokensPair&& tokens) { batch_id_to_token_pairs_[batch_id] = std::move(tokens); }
void ProcessingStorage::remove(int batch_id) { batch_id_to_token_pairs_.erase(batch_id); }
bool ProcessingStorage::batch_id_processing(int batch_id) { return batch_id_to_token_pairs_.count(batch_id) != 0; }
IdTokensPair& ProcessingStorage::get_token(int batch_id) { return batch_id_to_token_pairs_[batch_id]; }
int ProcessingStorage::size() const { return static_cast<int>(batch_id_to_token_pairs_.size()); }

void ProcessingStorage::move_to_finished(int batch_id, ItemStorage& item_storage) {
    auto it = batch_id_to_token_pairs_.find(batch_id);
    assert(it != batch_id_to_token_pairs_.end());
    item_storage.add_finished_item(std::move(it->second));
    batch_id_to_token_pairs_.erase(it);
}

void ProcessingStorage::move_to_new(int batch_id, ItemStorage& item_storage) {
    auto it = batch_id_to_token_pairs_.find(batch_id);
    assert(it != batch_id_to_token_pairs_.end());
    item_storage.add_new_item_to_head(std::move(it->second));
    batch_id_to_token_pairs_.erase(it);
}

void append_token_to_id_string_pair(IdTokensPair& id_string_pair, int to_add) { id_string_pair.second.push_back(to_add); }

bool is_done(ItemStorage& item_storage, ProcessingStorage& processing_storage) {
    return processing_storage.size() == 0 && item_storage.new_count() == 0;
}

// Upload the token prefixes of the rows that changed.  A few rows: one small copy each (only the prompt);
// many rows: one copy spanning first..last changed row -- every hipMemcpy costs ~5 us of host time, so
// hundreds of per-row copies would dominate an iteration (the reference copies all of inp[B, S] every time).
void upload_changed_rows(TensorInt& inp_device, TensorInt& inp_host, const std::vector<int>& slots,
                         const int* lengths, int n_sequence) {
    if (slots.empty()) return;
    if (slots.size() <= 4) {
        for (int slot : slots)
            inp_device.copy_range_from(inp_host, static_cast<size_t>(slot) * n_sequence, static_cast<size_t>(lengths[slot]));
        return;
    }
    const auto mm = std::minmax_element(slots.begin(), slots.end());
    const size_t first = static_cast<size_t>(*mm.first) * n_sequence;
    const size_t last = static_cast<size_t>(*mm.second) * n_sequence + static_cast<size_t>(lengths[*mm.second]);
    inp_device.copy_range_from(inp_host, first, last - first);
}

// ---- per-iteration host steps -------------------------------------------------------------------------
std::vector<int> process_decoder_result(const TensorInt& decoder_result_device, TensorInt& decoder_result_host,
                                        ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                        int n_sequence) {
    const int n_batch = static_cast<int>(decoder_result_host.shape()[0]);
    const int n_rounds = decoder_result_host.shape().size() == 2 ? static_cast<int>(decoder_result_host.shape()[1]) : 1;
    decoder_result_host.copy_from(decoder_result_device);  // the device -> host sync point of the iteration
    const int* tokens = decoder_result_host.data();

    std::vector<int> free_slots;
    int appended = 0;
    for (int b = 0; b < n_batch; ++b) {
        bool row_empty = false, row_done = false;
        for (int r = 0; r < n_rounds && !row_empty && !row_done; ++r) {
            const int tok = tokens[b * n_rounds + r];
            if (tok == EMPTY_ROW_TOKEN_ID) {
                row_empty = true;
                continue;
            }
            IdTokensPair& item = processing_storage.get_token(b);
            append_token_to_id_string_pair(item, tok);
            ++appended;
            row_done = tok == EOF_TOKEN_ID || static_cast<int>(item.second.size()) >= n_sequence;
        }
        if (row_done) processing_storage.move_to_finished(b, item_storage);
        if (row_done || row_empty) free_slots.push_back(b);
    }
    get_global_throughput_counter().add_record_if_recording(appended);
    return free_slots;
}

int insert_new_items(const std::vector<int>& finished_indices, TensorInt& inp_device, TensorInt& inp_host,
                     TensorInt& lengths_device, TensorInt& lengths_host, TensorInt& new_items_indices_device,
                     TensorInt& new_items_indices_host, ItemStorage& item_storage,
                     ProcessingStorage& processing_storage) {
    if (finished_indices.empty()) return 0;

    std::vector<IdTokensPair> fresh = item_storage.pop_new_items(static_cast<int>(finished_indices.size()));
    const int n_batch = static_cast<int>(inp_host.shape()[0]);
    const int n_sequence = static_cast<int>(inp_host.shape()[1]);
    int* inp = inp_host.data();
    int* lengths = lengths_host.data();
    int* new_idx = new_items_indices_host.data();

    // The reference reads inp and lengths back from the device here (src/item_storage.cpp:153-154).  The
    // device never writes inp, and the device length of an in-flight row always equals its host token
    // count (decoder: L+1, host: one token appended), so the host already holds both -- no D2H needed.
    for (int b = 0; b < n_batch; ++b)
        lengths[b] = processing_storage.batch_id_processing(b)
                         ? static_cast<int>(processing_storage.get_token(b).second.size()) : 0;

    std::vector<int> filled;
    for (size_t i = 0; i < finished_indices.size(); ++i) {
        const int slot = finished_indices[i];
        new_idx[i] = slot;
        if (i >= fresh.size()) {
            lengths[slot] = 0;  // queue ran dry: the slot stays empty
            continue;
        }
        std::vector<int>& toks = fresh[i].second;
        assert(static_cast<int>(toks.size()) + 1 <= n_sequence);
        lengths[slot] = static_cast<int>(toks.size());
        std::copy(toks.begin(), toks.end(), inp + static_cast<size_t>(slot) * n_sequence);
        filled.push_back(slot);
        processing_storage.put(slot, std::move(fresh[i]));
    }
    upload_changed_rows(inp_device, inp_host, filled, lengths, n_sequence);
    lengths_device.copy_from(lengths_host);
    new_items_indices_device.copy_from(new_items_indices_host);
    return static_cast<int>(fresh.size());
}
