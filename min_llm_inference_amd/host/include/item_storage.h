// Host side of continuous batching: the queue of waiting items, the finished items, and the map from
// batch slot to the item being generated (reference include/item_storage.h, src/item_storage.cpp).
#pragma once

#include <list>
#include <unordered_map>
#include <utility>
#include <vector>

#include "tensor.hpp"
#include "utils.h"

using IdTokensPair = std::pair<int, std::vector<int>>;  // (item id, prompt + generated token ids)

class Storage : public NonCopyableNonClonable {
public:
    Storage() = default;
    std::vector<IdTokensPair> pop_pairs(int size);  // up to `size` items from the front
    void add(IdTokensPair&&);
    void add_to_front(IdTokensPair&&);
    int size() const;
    int head_length() const;
    const IdTokensPair& get_top() const;
    const std::list<IdTokensPair>& get_data() const;

private:
    std::list<IdTokensPair> data_;
};

class ItemStorage : public NonCopyableNonClonable {
public:
    ItemStorage() = default;
    std::vector<IdTokensPair> pop_finished_items(int size);
    std::vector<IdTokensPair> pop_new_items(int size);
    const IdTokensPair& get_top() const;
    void add_finished_item(IdTokensPair&&);
    void add_new_item(IdTokensPair&&);
    void add_new_item_to_head(IdTokensPair&&);  // a preempted row goes back to the front of the queue
    int finish_count() const;
    int new_count() const;
    int head_length() const;  // token count of the item at the front of the queue
    const std::list<IdTokensPair>& get_finished_items() const;

private:
    Storage finished_items_;
    Storage new_items_;
};

class ProcessingStorage : public NonCopyableNonClonable {
public:
    ProcessingStorage() = default;
    void put(int batch_id, IdTokensPair&&);
    void remove(int batch_id);
    bool batch_id_processing(int batch_id);
    IdTokensPair& get_token(int batch_id);
    void move_to_new(int batch_id, ItemStorage& item_storage);
    void move_to_finished(int batch_id, ItemStorage& item_storage);
    int size() const;

private:
    std::unordered_map<int, IdTokensPair> batch_id_to_token_pairs_;
};

void append_token_to_id_string_pair(IdTokensPair& id_string_pair, int to_add);

// D2H copy of the decoder output, append tokens, detect finished rows (EOF or n_sequence tokens).
// Returns the slots that can take a new item (finished or empty), in slot order.
std::vector<int> process_decoder_result(const TensorInt& decoder_result_device, TensorInt& decoder_result_host,
                                        ItemStorage& item_storage, ProcessingStorage& processing_storage,
                                        int n_sequence);

// Contiguous engine: fill the given slots from the queue (length 0 when the queue runs dry), upload.
// Returns the number of newly inserted items.
int insert_new_items(const std::vector<int>& finished_indices, TensorInt& inp_device, TensorInt& inp_host,
                     TensorInt& lengths_device, TensorInt& lengths_host, TensorInt& new_items_indices_device,
                     TensorInt& new_items_indices_host, ItemStorage& item_storage,
                     ProcessingStorage& processing_storage);

bool is_done(ItemStorage& item_storage, ProcessingStorage& processing_storage);

// Extension: upload inp rows `slots` (lengths[slot] tokens each) with as few copies as possible.
void upload_changed_rows(TensorInt& inp_device, TensorInt& inp_host, const std::vector<int>& slots,
                         const int* lengths, int n_sequence);
