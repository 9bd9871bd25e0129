// CPU unit tests of the host scheduler (item_storage.cpp, paged_item_storage.cpp) over the malloc test double.
// Scenarios follow the reference's tests/item_storage_test.cpp:9-190 and tests/paged_item_storage_test.cpp:17-277
// (finish detection, slot refill with / without enough queued items, fill all slots, page return, +1 page growth,
// preempt the last row / the tail rows) with seeded inputs, plus the cases the rewrite added (row-range upload,
// length refresh vs the reference quirk, page-table width cap for n_forward_rounds > 1).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "constants.h"
#include "item_storage.h"
#include "paged_item_storage.h"
#include "throughput_counter.h"

static int g_failures = 0;
#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            std::printf("  CHECK failed: %s  (%s:%d)\n", #cond, __FILE__, __LINE__); \
            ++g_failures;                                                           \
        }                                                                           \
    } while (0)
#define CHECK_EQ(a, b)                                                                                  \
    do {                                                                                                \
        auto va = (a);                                                                                  \
        auto vb = (b);                                                                                  \
        if (!(va == vb)) {                                                                              \
            std::printf("  CHECK_EQ failed: %s == %s  (%lld vs %lld)  (%s:%d)\n", #a, #b, (long long)va, \
                        (long long)vb, __FILE__, __LINE__);                                             \
            ++g_failures;                                                                               \
        }                                                                                               \
    } while (0)

static std::mt19937 rng(20251004);
static int rnd(int lo, int hi) { return std::uniform_int_distribution<int>(lo, hi)(rng); }
static std::vector<int> rnd_tokens(int n) {
    std::vector<int> v(n);
    for (int& x : v) x = rnd(0, EOF_TOKEN_ID - 1);
    return v;
}
static std::vector<int> unique_sample(int lo, int hi, int n) {
    std::vector<int> all(hi - lo + 1);
    std::iota(all.begin(), all.end(), lo);
    std::shuffle(all.begin(), all.end(), rng);
    all.resize(n);
    return all;
}

struct Buffers {
    TensorInt inp_d, inp_h, len_d, len_h, idx_d, idx_h;
    Buffers(size_t B, size_t S)
        : inp_d({B, S}, DeviceType::DEVICE), inp_h({B, S}, DeviceType::HOST), len_d({B}, DeviceType::DEVICE),
          len_h({B}, DeviceType::HOST), idx_d({B}, DeviceType::DEVICE), idx_h({B}, DeviceType::HOST) {
        std::fill(inp_h.data(), inp_h.data() + B * S, -7);
        std::fill(inp_d.data(), inp_d.data() + B * S, -7);
        std::fill(len_h.data(), len_h.data() + B, 99);
        std::fill(len_d.data(), len_d.data() + B, 99);
    }
};

struct PagedWorld {
    PagedAttentionsManager pages;
    MemoryBlockManager pool;
    ProcessingStorage processing;
    ItemStorage items;
    std::vector<IdTokensPair> tokens;
    PagedWorld(size_t B, size_t S, size_t D, int n_blocks, const std::vector<int>& lengths)
        : pages(B, S, D), pool(n_blocks, PAGE_BLOCK_SIZE * 3 * D) {
        for (size_t i = 0; i < lengths.size(); ++i) {
            tokens.emplace_back((int)i, rnd_tokens(lengths[i]));
            items.add_new_item(IdTokensPair(tokens.back()));
        }
    }
    std::vector<int> insert(Buffers& b, int rounds = 1) {
        return insert_new_items(b.inp_d, b.inp_h, b.len_d, b.len_h, b.idx_d, b.idx_h, items, processing, pool, pages, rounds);
    }
};

static void decode_step(PagedWorld& w, size_t B, size_t S, const std::vector<int>& result, std::vector<int>* finished,
                        int rounds = 1) {
    TensorInt res_d({B}, DeviceType::DEVICE), res_h({B}, DeviceType::HOST);
    std::copy(result.begin(), result.end(), res_d.data());
    *finished = process_decoder_result(res_d, res_h, w.items, w.processing, (int)S);
    allocate_or_free_memory_blocks_if_needed(w.pages, w.pool, w.processing, w.items, *finished, rounds);
}

// ---- contiguous scheduler -------------------------------------------------------------------------------
static void test_process_decoder_result() {
    const int B = rnd(40, 90), S = rnd(20, 60);
    ItemStorage items;
    ProcessingStorage processing;
    std::vector<int> result(B);
    std::vector<int> expect_free;
    int n_empty = 0, n_done = 0;
    for (int b = 0; b < B; ++b) {
        const int kind = rnd(0, 3);  // 0 empty slot, 1 EOF, 2 reaches n_sequence, 3 keeps going
        if (kind == 0) {
            result[b] = EMPTY_ROW_TOKEN_ID;
            ++n_empty;
            expect_free.push_back(b);
            continue;
        }
        const int len = kind == 2 ? S - 1 : rnd(1, S - 3);
        processing.put(b, std::make_pair(b, rnd_tokens(len)));
        result[b] = kind == 1 ? EOF_TOKEN_ID : rnd(0, EOF_TOKEN_ID - 1);
        if (kind == 1 || kind == 2) {
            ++n_done;
            expect_free.push_back(b);
        }
    }
    TensorInt res_d({(size_t)B}, DeviceType::DEVICE), res_h({(size_t)B}, DeviceType::HOST);
    std::copy(result.begin(), result.end(), res_d.data());
    get_global_throughput_counter().reset();
    get_global_throughput_counter().start_record();
    std::vector<int> free_slots = process_decoder_result(res_d, res_h, items, processing, S);
    CHECK(free_slots == expect_free);
    CHECK_EQ(items.finish_count(), n_done);
    CHECK_EQ(items.finish_count() + processing.size() + n_empty, B);
    CHECK_EQ(get_global_throughput_counter().total_tokens(), (long long)(B - n_empty));
    for (const auto& it : items.get_finished_items()) CHECK(it.second.back() == result[it.first]);
}

static void test_insert_new_items(bool enough) {
    const int B = rnd(30, 80), S = rnd(24, 64);
    ItemStorage items;
    ProcessingStorage processing;
    Buffers buf(B, S);
    for (int b = 0; b < B; ++b) processing.put(b, std::make_pair(b, rnd_tokens(rnd(1, S - 2))));
    const int n_free = rnd(3, B - 2);
    std::vector<int> free_slots = unique_sample(0, B - 1, n_free);
    std::sort(free_slots.begin(), free_slots.end());
    for (int s : free_slots) processing.remove(s);
    const int n_queued = enough ? n_free + rnd(0, 5) : rnd(1, n_free - 1);
    for (int i = 0; i < n_queued; ++i) items.add_new_item(std::make_pair(B + i, rnd_tokens(rnd(1, S - 1))));
    const int n_new = insert_new_items(free_slots, buf.inp_d, buf.inp_h, buf.len_d, buf.len_h, buf.idx_d, buf.idx_h, items, processing);
    CHECK_EQ(n_new, std::min(n_free, n_queued));
    const int* len_d = buf.len_d.data();
    const int* inp_d = buf.inp_d.data();
    const int* idx_d = buf.idx_d.data();
    for (int i = 0; i < n_free; ++i) {
        const int slot = free_slots[i];
        CHECK_EQ(idx_d[i], slot);
        if (i < n_queued) {
            CHECK_EQ(processing.get_token(slot).first, B + i);  // queue order is preserved
            const auto& toks = processing.get_token(slot).second;
            CHECK_EQ(len_d[slot], (int)toks.size());
            for (size_t j = 0; j < toks.size(); ++j) CHECK_EQ(inp_d[slot * S + j], toks[j]);  // row-range upload
            CHECK_EQ(inp_d[slot * S + toks.size()], -7);                                      // and nothing past it
        } else {
            CHECK(!processing.batch_id_processing(slot));
            CHECK_EQ(len_d[slot], 0);
        }
    }
    // in-flight rows: device length == host token count (refreshed from host truth, no D2H needed)
    std::set<int> freed(free_slots.begin(), free_slots.end());
    for (int b = 0; b < B; ++b)
        if (!freed.count(b)) CHECK_EQ(len_d[b], (int)processing.get_token(b).second.size());
    CHECK_EQ(insert_new_items({}, buf.inp_d, buf.inp_h, buf.len_d, buf.len_h, buf.idx_d, buf.idx_h, items, processing), 0);
}

// ---- paged scheduler --------------------------------------------------------------------------------------
static void test_paged_insert_all() {
    const size_t B = rnd(64, 200), S = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS * rnd(2, 5), D = 4 * rnd(8, 16);
    std::vector<int> lengths(2 * B);
    for (int& l : lengths) l = rnd(1, PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS - 1);
    PagedWorld w(B, S, D, DEFAULT_INIT_NUM_BLOCKS * B, lengths);
    Buffers buf(B, S);
    std::vector<int> slots = w.insert(buf);
    CHECK_EQ(slots.size(), B);
    for (size_t i = 0; i < B; ++i) CHECK_EQ(slots[i], (int)i);
    CHECK_EQ(w.items.new_count(), (int)B);
    CHECK_EQ(w.pool.free_blocks_size(), 0);
    float** table = w.pages.get_page_table_device().data();
    std::set<float*> seen;
    for (size_t b = 0; b < B; ++b) {
        CHECK_EQ(buf.len_d.data()[b], lengths[b]);
        CHECK_EQ(buf.idx_d.data()[b], (int)b);
        for (int j = 0; j < lengths[b]; ++j) CHECK_EQ(buf.inp_d.data()[b * S + j], w.tokens[b].second[j]);
        for (int p = 0; p < DEFAULT_INIT_NUM_BLOCKS; ++p) seen.insert(table[b * (S / PAGE_BLOCK_SIZE) + p]);
    }
    CHECK_EQ(seen.size(), B * DEFAULT_INIT_NUM_BLOCKS);  // every slot owns distinct pages, flushed to the device table
    CHECK(w.insert(buf).empty());                        // nothing free: no admission, no upload
}

static void test_paged_return_blocks() {
    const size_t B = rnd(64, 200), S = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS * rnd(2, 5), D = 32;
    std::vector<int> lengths(2 * B);
    for (int& l : lengths) l = rnd(1, PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS - 2);
    PagedWorld w(B, S, D, DEFAULT_INIT_NUM_BLOCKS * B, lengths);
    Buffers buf(B, S);
    w.insert(buf);
    CHECK_EQ(w.pool.free_blocks_size(), 0);
    const int n_fin = rnd(2, (int)B - 10);
    std::vector<int> fin = unique_sample(0, (int)B - 1, n_fin);
    std::vector<int> result(B);
    for (int& r : result) r = rnd(0, EOF_TOKEN_ID - 1);
    for (int f : fin) result[f] = EOF_TOKEN_ID;
    std::vector<int> finished;
    decode_step(w, B, S, result, &finished);
    CHECK_EQ(w.pool.free_blocks_size(), n_fin * DEFAULT_INIT_NUM_BLOCKS);
    CHECK_EQ(w.items.finish_count(), n_fin);
    std::vector<int> slots = w.insert(buf);
    CHECK_EQ((int)slots.size(), n_fin);
    std::sort(fin.begin(), fin.end());
    CHECK(slots == fin);
    // rows still in flight keep their true (grown) length on the device -- the reference would reset them here
    std::set<int> fs(fin.begin(), fin.end());
    for (size_t b = 0; b < B; ++b)
        if (!fs.count((int)b)) CHECK_EQ(buf.len_d.data()[b], lengths[b] + 1);
}

static void test_paged_length_reset_quirk() {
    const size_t B = 16, S = 128, D = 16;
    std::vector<int> lengths(B + 1, 10);
    PagedWorld w(B, S, D, 8 * B, lengths);
    Buffers buf(B, S);
    w.insert(buf);
    std::vector<int> result(B, 5), finished;
    result[3] = EOF_TOKEN_ID;
    decode_step(w, B, S, result, &finished);
    // a second engine in the same process, stepped in between: the switch belongs to ONE manager (a
    // process-wide flag let one engine's start() change another engine's uploads mid-run)
    PagedWorld other(B, S, D, 8 * B, lengths);
    Buffers other_buf(B, S);
    other.insert(other_buf);
    std::vector<int> other_finished;
    decode_step(other, B, S, result, &other_finished);
    w.pages.set_length_reset_quirk(true);
    CHECK(!other.pages.length_reset_quirk());
    w.insert(buf);
    CHECK_EQ(buf.len_d.data()[0], 10);  // stale insertion-time length, as src/paged_item_storage.cpp:110-118 uploads it
    other.insert(other_buf);
    CHECK_EQ(other_buf.len_d.data()[0], 11);  // the other engine still uploads the truth
    w.pages.set_length_reset_quirk(false);
    // the process-wide switch is the default of managers built afterwards, nothing more
    set_reference_length_reset_quirk(true);
    CHECK(!w.pages.length_reset_quirk());
    {
        PagedAttentionsManager later(B, S, D);
        CHECK(later.length_reset_quirk());
    }
    set_reference_length_reset_quirk(false);
    std::vector<int> result2(B, 5);
    result2[4] = EOF_TOKEN_ID;
    decode_step(w, B, S, result2, &finished);
    w.insert(buf);
    CHECK_EQ(buf.len_d.data()[0], 12);  // host truth: prompt + two generated tokens
}

static void test_paged_allocate_more() {
    const size_t B = 2 * rnd(40, 100), S = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS * rnd(2, 5), D = 32;
    const int full = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS;
    std::vector<int> lengths(B / 2);
    for (int& l : lengths) l = rnd(1, full - 2);
    const int n_grow = rnd(2, (int)B / 2);
    for (int i : unique_sample(0, (int)B / 2 - 1, n_grow)) lengths[i] = full - 1;
    const int n_blocks = DEFAULT_INIT_NUM_BLOCKS * B;
    PagedWorld w(B, S, D, n_blocks, lengths);
    Buffers buf(B, S);
    w.insert(buf);
    CHECK_EQ(w.pool.free_blocks_size(), n_blocks - (int)B / 2 * DEFAULT_INIT_NUM_BLOCKS);
    std::vector<int> result(B, EMPTY_ROW_TOKEN_ID), finished;
    for (size_t b = 0; b < B / 2; ++b) result[b] = rnd(0, EOF_TOKEN_ID - 1);
    decode_step(w, B, S, result, &finished);
    CHECK_EQ((int)finished.size(), (int)B / 2);  // only the empty upper half reports "free"
    CHECK_EQ(w.pool.free_blocks_size(), n_blocks - (int)B / 2 * DEFAULT_INIT_NUM_BLOCKS - n_grow);
    // the new page landed at index 4 of exactly the rows that crossed the boundary
    float** host_table_after_flush = nullptr;
    w.pages.maybe_flush_changes();
    host_table_after_flush = w.pages.get_page_table_device().data();
    for (size_t b = 0; b < B / 2; ++b) {
        bool grew = lengths[b] == full - 1;
        const auto& used = w.pages.get_used_block_list();
        auto it = std::find_if(used.begin(), used.end(), [&](const BatchIdMemoryBlocksPair& r) { return r.first == (int)b; });
        CHECK_EQ((int)it->second.size(), DEFAULT_INIT_NUM_BLOCKS + (grew ? 1 : 0));
        if (grew) CHECK(host_table_after_flush[b * (S / PAGE_BLOCK_SIZE) + DEFAULT_INIT_NUM_BLOCKS] == it->second.front());
    }
}

static void test_paged_preempt_last() {
    const size_t B = rnd(64, 200), S = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS * rnd(2, 5), D = 32;
    const int full = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS;
    std::vector<int> lengths(2 * B);
    for (int& l : lengths) l = rnd(1, full - 2);
    lengths[B - 1] = full - 1;  // only the last admitted row needs a page, and none is free
    PagedWorld w(B, S, D, DEFAULT_INIT_NUM_BLOCKS * B, lengths);
    Buffers buf(B, S);
    w.insert(buf);
    CHECK_EQ(w.pool.free_blocks_size(), 0);
    std::vector<int> result(B), finished;
    for (int& r : result) r = rnd(0, EOF_TOKEN_ID - 1);
    decode_step(w, B, S, result, &finished);
    CHECK_EQ(w.pool.free_blocks_size(), DEFAULT_INIT_NUM_BLOCKS);
    CHECK_EQ(w.items.new_count(), (int)B + 1);   // the preempted row is back in the queue...
    CHECK_EQ(w.items.head_length(), full);       // ...at its head, with the generated token kept
    CHECK_EQ(w.items.pop_new_items(1)[0].second.back(), result[B - 1]);
    CHECK(!w.processing.batch_id_processing((int)B - 1));
}

static void test_paged_preempt_tail() {
    const size_t B = rnd(64, 200), S = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS * rnd(2, 5), D = 32;
    const int full = PAGE_BLOCK_SIZE * DEFAULT_INIT_NUM_BLOCKS;
    std::vector<int> lengths(B - 1);
    for (int& l : lengths) l = rnd(1, full - 2);
    const int to_fill = rnd(DEFAULT_INIT_NUM_BLOCKS + 1, DEFAULT_INIT_NUM_BLOCKS * 5 + 1);
    const int to_free = ceil_div(to_fill, DEFAULT_INIT_NUM_BLOCKS) - 1;  // one row's worth of pages is free already
    for (int i : unique_sample(0, (int)B - 2 - to_free, to_fill)) lengths[i] = full - 1;
    PagedWorld w(B, S, D, DEFAULT_INIT_NUM_BLOCKS * B, lengths);
    Buffers buf(B, S);
    w.insert(buf);
    CHECK_EQ(w.pool.free_blocks_size(), DEFAULT_INIT_NUM_BLOCKS);
    std::vector<int> result(B, EMPTY_ROW_TOKEN_ID), finished;
    for (size_t b = 0; b + 1 < B; ++b) result[b] = rnd(0, EOF_TOKEN_ID - 1);
    decode_step(w, B, S, result, &finished);
    CHECK_EQ(w.pool.free_blocks_size(), DEFAULT_INIT_NUM_BLOCKS * to_free + DEFAULT_INIT_NUM_BLOCKS - to_fill);
    CHECK_EQ(w.items.new_count(), to_free);  // the most recently admitted rows went back to the queue
    for (int k = 0; k < to_free; ++k) CHECK(!w.processing.batch_id_processing((int)B - 2 - k));
}

static void test_paged_width_cap_multi_round() {
    // n_forward_rounds > 1 near n_sequence: the row must NOT be given page index == table width
    const size_t B = 4, S = 64, D = 16;
    const int rounds = 4;
    std::vector<int> lengths = {61, 10, 10, 10};
    PagedWorld w(B, S, D, 64, lengths);
    Buffers buf(B, S);
    w.insert(buf, rounds);
    float** table = w.pages.get_page_table_device().data();
    std::vector<float*> row1_before(table + 4, table + 8);
    TensorInt res_d({B, (size_t)rounds}, DeviceType::DEVICE), res_h({B, (size_t)rounds}, DeviceType::HOST);
    for (size_t i = 0; i < B * rounds; ++i) res_d.data()[i] = 7;
    res_d.data()[0 * rounds + 1] = 9;  // row 0: 61 -> 62, 63 tokens ...
    std::vector<int> finished = process_decoder_result(res_d, res_h, w.items, w.processing, (int)S);
    allocate_or_free_memory_blocks_if_needed(w.pages, w.pool, w.processing, w.items, finished, rounds);
    w.pages.maybe_flush_changes();
    for (int p = 0; p < 4; ++p) CHECK(table[4 + p] == row1_before[p]);  // row 1's entries untouched
    for (const auto& r : w.pages.get_used_block_list()) CHECK((int)r.second.size() <= (int)(S / PAGE_BLOCK_SIZE));
}

static void test_pool_exhaustion_throws() {
    MemoryBlockManager pool(3, 48);
    bool threw = false;
    try {
        pool.pop_free_blocks(4);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
    auto two = pool.pop_free_blocks(2);
    CHECK_EQ(pool.free_blocks_size(), 1);
    pool.return_free_blocks(std::move(two));
    CHECK_EQ(pool.free_blocks_size(), 3);
}

static void test_tensor_semantics() {
    TensorInt a({2, 3}, DeviceType::HOST), c({2, 3}, DeviceType::DEVICE);
    TensorInt alias = a;  // copy = alias of the same storage (reference tensor.hpp:101-102)
    a.data()[4] = 42;
    CHECK_EQ(alias.data()[4], 42);
    CHECK_EQ(a.get_total_size(), (size_t)6);
    c.copy_from(a);
    CHECK_EQ(c.data()[4], 42);
    TensorInt wrong({7}, DeviceType::HOST);
    bool threw = false;
    try {
        wrong.copy_from(a);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
    TensorInt other({2, 3}, DeviceType::HOST, TensorDataType::ASYNC_ALLOCATE);
    threw = false;
    try {
        other.copy_from(a);  // mixing allocation flavours is an error in the reference too
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);
}

int main() {
    struct Case { const char* name; std::function<void()> fn; int reps; };
    const Case cases[] = {
        {"process_decoder_result", test_process_decoder_result, 20},
        {"insert_new_items (enough queued)", [] { test_insert_new_items(true); }, 20},
        {"insert_new_items (queue runs dry)", [] { test_insert_new_items(false); }, 20},
        {"paged: fill every slot", test_paged_insert_all, 5},
        {"paged: finished rows return pages", test_paged_return_blocks, 5},
        {"paged: reference length-reset quirk switch", test_paged_length_reset_quirk, 1},
        {"paged: +1 page growth", test_paged_allocate_more, 5},
        {"paged: preempt the last row", test_paged_preempt_last, 5},
        {"paged: preempt tail rows", test_paged_preempt_tail, 5},
        {"paged: page-table width cap (n_forward_rounds > 1)", test_paged_width_cap_multi_round, 1},
        {"pool exhaustion throws", test_pool_exhaustion_throws, 1},
        {"Tensor alias / copy_from semantics", test_tensor_semantics, 1},
    };
    for (const Case& c : cases) {
        const int before = g_failures;
        for (int r = 0; r < c.reps; ++r) c.fn();
        std::printf("[%s] %s\n", g_failures == before ? " OK " : "FAIL", c.name);
    }
    std::printf("%d failure(s)\n", g_failures);
    return g_failures ? 1 : 0;
}
