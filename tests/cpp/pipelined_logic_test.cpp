// CPU test of the pipelined engine loop's scheduling logic (min_llm_inference_amd/host/src/pipelined_engine.cpp) over
// the malloc test double, under ASan + UBSan.  A FAKE model stands in for the GPU forward: it keeps the decoder's
// contract on the "device" tensors (token from a running hash of the row's whole token prefix, length L -> L + 1,
// or 0 when the row finishes on EOF / max length; EMPTY for rows of length 0) and checks, every step, that each
// live row has a page for every position it reads and for the slot it writes next.  Greedy decoding by such a model
// is a pure function of the prompt, so the sequential loop (the reference's order, rebuilt here from the scheduler
// primitives) and the pipelined loop must finish the same items with the same tokens -- under roomy pools, tight
// pools (constant preemption, in-flight tokens dropped and regenerated) and EOF-heavy vocabularies.
#include <cstdio>
#include <cstdint>
#include <map>
#include <random>
#include <stdexcept>
#include <vector>

#include "constants.h"
#include "pipelined_engine.h"
#include "throughput_counter.h"

static int g_failures = 0;
#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            std::printf("  CHECK failed: %s  (%s:%d)\n", #cond, __FILE__, __LINE__); \
            ++g_failures;                                                           \
        }                                                                           \
    } while (0)

struct FakeModel {
    int B, S, eof_bias;
    std::vector<uint64_t> h;      // running hash per slot (the "KV cache" of the fake)
    PagedAttentionsManager* pages;
    long long launches = 0;
    int missing_pages = 0;
    int rounds = 1;   // decode rounds per forward: result is [B, rounds], a finished row reports EMPTY in later rounds

    static uint64_t mix(uint64_t h, uint64_t x) {
        h ^= x + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        return h * 0xff51afd7ed558ccdull;
    }
    int token_of(uint64_t hh) const {
        // eof_bias in [0, 100]: percentage of steps that emit EOF
        const int r = static_cast<int>((hh >> 17) % 100);
        if (r < eof_bias) return EOF_TOKEN_ID;
        return static_cast<int>((hh >> 33) % EOF_TOKEN_ID);
    }
    void forward(const TensorInt& inp, TensorInt& lengths, const TensorInt& new_idx, TensorInt& result, int n_new) {
        ++launches;
        const int* in = inp.data();
        int* len = lengths.data();
        int* res = result.data();
        float** table = pages->get_page_table_device().data();
        const int width = S / PAGE_BLOCK_SIZE;
        for (int i = 0; i < n_new; ++i) {  // prefill: hash of the whole prompt
            const int b = new_idx.data()[i];
            uint64_t hh = 0x1234;
            for (int s = 0; s < len[b]; ++s) hh = mix(hh, static_cast<uint64_t>(in[b * S + s]));
            h[b] = hh;
        }
        for (int r = 0; r < rounds; ++r) {
            for (int b = 0; b < B; ++b) {
                const int L = len[b];
                if (L <= 0) {
                    res[b * rounds + r] = EMPTY_ROW_TOKEN_ID;
                    continue;
                }
                for (int s = 0; s <= L && s < S; s += PAGE_BLOCK_SIZE)   // pages of positions 0 .. L (L = next write)
                    if (table[b * width + s / PAGE_BLOCK_SIZE] == nullptr) ++missing_pages;
                if (L < S && table[b * width + L / PAGE_BLOCK_SIZE] == nullptr) ++missing_pages;
                const int tok = token_of(h[b]);
                res[b * rounds + r] = tok;
                h[b] = mix(h[b], static_cast<uint64_t>(tok));
                len[b] = (tok == EOF_TOKEN_ID || L + 1 >= S) ? 0 : L + 1;
            }
        }
    }
};

struct World {
    ItemStorage items;
    ProcessingStorage processing;
    MemoryBlockManager pool;
    PagedAttentionsManager pages;
    World(size_t B, size_t S, int n_blocks) : pool(n_blocks, PAGE_BLOCK_SIZE * 3 * 4), pages(B, S, 4) {
        float** t = pages.get_page_table_device().data();   // the double hands out uninitialised memory
        for (size_t i = 0; i < B * (S / PAGE_BLOCK_SIZE); ++i) t[i] = nullptr;
    }
};

static std::map<int, std::vector<int>> collect(const ItemStorage& s) {
    std::map<int, std::vector<int>> out;
    for (const auto& it : s.get_finished_items()) out[it.first] = it.second;
    return out;
}

// the reference's loop order (src/inferencer.cpp:43-122) from the scheduler primitives
static long long run_sequential(World& w, FakeModel& model, size_t B, size_t S, int R = 1) {
    TensorInt inp_d({B, S}, DeviceType::DEVICE), inp_h({B, S}, DeviceType::HOST);
    TensorInt len_d({B}, DeviceType::DEVICE), len_h({B}, DeviceType::HOST);
    TensorInt idx_d({B}, DeviceType::DEVICE), idx_h({B}, DeviceType::HOST);
    TensorInt res_d({B, (size_t)R}, DeviceType::DEVICE), res_h({B, (size_t)R}, DeviceType::HOST);
    for (size_t b = 0; b < B; ++b) len_h.data()[b] = len_d.data()[b] = 0;
    std::vector<int> fresh = insert_new_items(inp_d, inp_h, len_d, len_h, idx_d, idx_h, w.items, w.processing, w.pool, w.pages, R);
    long long steps = 0;
    while (!is_done(w.items, w.processing)) {
        model.forward(inp_d, len_d, idx_d, res_d, static_cast<int>(fresh.size()));
        std::vector<int> finished = process_decoder_result(res_d, res_h, w.items, w.processing, static_cast<int>(S));
        allocate_or_free_memory_blocks_if_needed(w.pages, w.pool, w.processing, w.items, finished, R);
        fresh = insert_new_items(inp_d, inp_h, len_d, len_h, idx_d, idx_h, w.items, w.processing, w.pool, w.pages, R);
        if (w.processing.size() == 0 && w.items.new_count() > 0) return -1;   // inferencer.cpp: throw_if_stuck
        if (++steps > 1000000) break;
    }
    return steps;
}

// Pools SMALLER than the page-table width: the look-ahead of the pipelined loop (tokens + 2 R positions, asked
// for before result(step) is known) must not cost the last row in flight its place when the row is in fact finishing in
// forward(step).  Whatever the sequential loop does with the workload -- finish it, or report the pool as too small --
// the pipelined loop must do too, with the same tokens.
static void run_tight_case(unsigned seed, size_t B, size_t S, int n_blocks, int n_items, int max_prompt, int eof_bias, int rounds) {
    std::mt19937 rng(seed);
    std::vector<IdTokensPair> items;
    for (int i = 0; i < n_items; ++i) {
        std::vector<int> toks(1 + rng() % max_prompt);
        for (int& t : toks) t = static_cast<int>(rng() % EOF_TOKEN_ID);
        items.emplace_back(i, toks);
    }
    std::map<int, std::vector<int>> seq, pip;
    long long seq_steps;
    {
        World w(B, S, n_blocks);
        for (const auto& it : items) w.items.add_new_item(IdTokensPair(it));
        FakeModel model{(int)B, (int)S, eof_bias, std::vector<uint64_t>(B, 0), &w.pages};
        model.rounds = rounds;
        get_global_throughput_counter().reset();
        get_global_throughput_counter().start_record();
        seq_steps = run_sequential(w, model, B, S, rounds);
        seq = collect(w.items);
        CHECK(model.missing_pages == 0);
    }
    bool threw = false;
    int missing = 0;
    {
        World w(B, S, n_blocks);
        for (const auto& it : items) w.items.add_new_item(IdTokensPair(it));
        FakeModel model{(int)B, (int)S, eof_bias, std::vector<uint64_t>(B, 0), &w.pages};
        model.rounds = rounds;
        get_global_throughput_counter().reset();
        try {
            run_paged_engine_pipelined(w.items, w.processing, w.pool, w.pages, B, S,
                                       [&](const TensorInt& inp, TensorInt& len, const TensorInt& idx, TensorInt& res, int n_new) {
                                           if (model.launches > 200000) throw std::logic_error("no progress");
                                           model.forward(inp, len, idx, res, n_new);
                                       }, rounds);
        } catch (const std::runtime_error&) {
            threw = true;
        }
        pip = collect(w.items);
        missing = model.missing_pages;
    }
    CHECK(missing == 0);
    CHECK(threw == (seq_steps < 0));
    int different = 0;
    if (seq_steps >= 0) {
        CHECK((int)pip.size() == n_items);
        for (const auto& kv : seq) different += pip[kv.first] != kv.second;
        CHECK(different == 0);
    }
    std::printf("%s tight seed %u: B=%zu S=%zu blocks=%d (width %zu) items=%d eof=%d%% rounds=%d  sequential %s, pipelined %s\n",
                missing == 0 && threw == (seq_steps < 0) && different == 0 ? "[ OK ]" : "[FAIL]", seed, B, S, n_blocks,
                S / PAGE_BLOCK_SIZE, n_items, eof_bias, rounds, seq_steps < 0 ? "stuck" : "finished", threw ? "reported" : "finished");
}

static void run_case(unsigned seed, size_t B, size_t S, int n_blocks, int n_items, int max_prompt, int eof_bias,
                     int rounds = 1) {
    std::mt19937 rng(seed);
    std::vector<IdTokensPair> items;
    for (int i = 0; i < n_items; ++i) {
        std::vector<int> toks(1 + rng() % max_prompt);
        for (int& t : toks) t = static_cast<int>(rng() % EOF_TOKEN_ID);
        items.emplace_back(i, toks);
    }
    std::map<int, std::vector<int>> seq, pip;
    int missing_seq = 0, missing_pip = 0;
    long long appended_seq = 0, appended_pip = 0;
    {
        World w(B, S, n_blocks);
        for (const auto& it : items) w.items.add_new_item(IdTokensPair(it));
        FakeModel model{(int)B, (int)S, eof_bias, std::vector<uint64_t>(B, 0), &w.pages};
        model.rounds = rounds;
        get_global_throughput_counter().reset();
        get_global_throughput_counter().start_record();
        run_sequential(w, model, B, S, rounds);
        appended_seq = get_global_throughput_counter().total_tokens();
        seq = collect(w.items);
        missing_seq = model.missing_pages;
        CHECK(w.pool.free_blocks_size() == n_blocks);   // every page came back
    }
    {
        World w(B, S, n_blocks);
        for (const auto& it : items) w.items.add_new_item(IdTokensPair(it));
        FakeModel model{(int)B, (int)S, eof_bias, std::vector<uint64_t>(B, 0), &w.pages};
        model.rounds = rounds;
        get_global_throughput_counter().reset();
        run_paged_engine_pipelined(w.items, w.processing, w.pool, w.pages, B, S,
                                   [&](const TensorInt& inp, TensorInt& len, const TensorInt& idx, TensorInt& res, int n_new) {
                                       model.forward(inp, len, idx, res, n_new);
                                   }, rounds);
        appended_pip = get_global_throughput_counter().total_tokens();
        pip = collect(w.items);
        missing_pip = model.missing_pages;
        CHECK(w.pool.free_blocks_size() == n_blocks);
        CHECK(w.processing.size() == 0 && w.items.new_count() == 0);
    }
    CHECK(missing_seq == 0);
    CHECK(missing_pip == 0);
    CHECK((int)seq.size() == n_items);
    CHECK((int)pip.size() == n_items);
    CHECK(appended_seq == appended_pip);
    int different = 0;
    for (const auto& kv : seq) different += pip[kv.first] != kv.second;
    CHECK(different == 0);
    std::printf("%s seed %u: B=%zu S=%zu blocks=%d items=%d eof=%d%% rounds=%d  tokens %lld  (items differing: %d)\n",
                different == 0 && missing_pip == 0 ? "[ OK ]" : "[FAIL]", seed, B, S, n_blocks, n_items, eof_bias, rounds,
                appended_pip, different);
}

int main() {
    std::mt19937 rng(77);
    for (unsigned seed = 0; seed < 40; ++seed) {
        const size_t B = 1 + rng() % 24;
        const size_t S = 16 * (2 + rng() % 9);
        const int width = static_cast<int>(S / 16);
        const int n_blocks = std::max<int>(width + DEFAULT_INIT_NUM_BLOCKS, (1 + rng() % 8) * static_cast<int>(B));
        const int n_items = 1 + rng() % (3 * B + 3);
        const int max_prompt = 1 + rng() % (S - 2);
        const int eof_bias = (seed % 3 == 0) ? 0 : static_cast<int>(rng() % 12);
        run_case(1000 + seed, B, S, n_blocks, n_items, max_prompt, eof_bias);
    }
    // several decode rounds per forward: up to R tokens of a row in flight, rows finishing in the middle of a forward,
    // preemption dropping up to R generated tokens at once
    for (unsigned seed = 0; seed < 30; ++seed) {
        const size_t B = 1 + rng() % 24;
        const size_t S = 16 * (2 + rng() % 9);
        const int width = static_cast<int>(S / 16);
        const int n_blocks = std::max<int>(width + DEFAULT_INIT_NUM_BLOCKS, (1 + rng() % 8) * static_cast<int>(B));
        const int n_items = 1 + rng() % (3 * B + 3);
        const int rounds = 2 + rng() % 7;   // 2 .. 8
        const int max_prompt = 1 + rng() % (S - 2 - rounds);
        const int eof_bias = (seed % 3 == 0) ? 0 : static_cast<int>(rng() % 12);
        run_case(2000 + seed, B, S, n_blocks, n_items, max_prompt, eof_bias, rounds);
    }
    {
        int finished_cases = 0;
        for (unsigned seed = 0; seed < 60; ++seed) {
            const size_t B = 1 + rng() % 6;
            const size_t S = 16 * (6 + rng() % 11);          // width 6 .. 16
            const int width = static_cast<int>(S / 16);
            const int n_blocks = DEFAULT_INIT_NUM_BLOCKS + rng() % (width - DEFAULT_INIT_NUM_BLOCKS);   // [4, width)
            const int rounds = (seed % 3 == 0) ? 1 : 2 + rng() % 7;
            const int n_items = 1 + rng() % 8;
            const int room = n_blocks * 16 - rounds - 2;
            const int max_prompt = 1 + rng() % std::max(1, std::min<int>(room, S - 2 - rounds) / 2);
            const int eof_bias = 2 + rng() % 10;            // rows end on EOF somewhere inside the pool, or outgrow it
            const int before = g_failures;
            run_tight_case(3000 + seed, B, S, n_blocks, n_items, max_prompt, eof_bias, rounds);
            finished_cases += g_failures == before;
        }
        CHECK(finished_cases > 0);
    }
    {   // the row's last token lands exactly on a page boundary and is EOF: one row, a pool of exactly the pages it uses
        for (int rounds : {1, 2, 4, 8}) {
            World w(1, 160, DEFAULT_INIT_NUM_BLOCKS);
            // find a prompt whose greedy continuation under the fake model emits EOF as token number 64 (4 pages full)
            bool found = false;
            std::vector<int> prompt;
            for (int first = 0; first < 200000 && !found; ++first) {
                FakeModel probe{1, 160, 3, std::vector<uint64_t>(1, 0), nullptr};
                uint64_t hh = FakeModel::mix(0x1234, (uint64_t)(first % EOF_TOKEN_ID));
                hh = FakeModel::mix(hh, (uint64_t)(first / EOF_TOKEN_ID));
                int n = 2;
                while (n < 64) {
                    const int tok = probe.token_of(hh);
                    if (tok == EOF_TOKEN_ID) break;
                    hh = FakeModel::mix(hh, (uint64_t)tok);
                    ++n;
                }
                if (n == 63 && probe.token_of(hh) == EOF_TOKEN_ID) {
                    found = true;
                    prompt = {first % EOF_TOKEN_ID, first / EOF_TOKEN_ID};
                    w.items.add_new_item(IdTokensPair(0, prompt));
                }
            }
            CHECK(found);
            if (!found) continue;
            // what the reference's loop order does with it: with R rounds it asks for tokens + R positions after every
            // result, so it completes the row only where 62 % R == 0 (the row's last forward then starts at 64 - R tokens)
            long long seq_steps;
            {
                World ws(1, 160, DEFAULT_INIT_NUM_BLOCKS);
                ws.items.add_new_item(IdTokensPair(0, prompt));
                FakeModel ms{1, 160, 3, std::vector<uint64_t>(1, 0), &ws.pages};
                ms.rounds = rounds;
                seq_steps = run_sequential(ws, ms, 1, 160, rounds);
            }
            FakeModel model{1, 160, 3, std::vector<uint64_t>(1, 0), &w.pages};
            model.rounds = rounds;
            bool threw = false;
            try {
                run_paged_engine_pipelined(w.items, w.processing, w.pool, w.pages, 1, 160,
                                           [&](const TensorInt& inp, TensorInt& len, const TensorInt& idx, TensorInt& res, int n_new) {
                                               model.forward(inp, len, idx, res, n_new);
                                           }, rounds);
            } catch (const std::runtime_error&) {
                threw = true;
            }
            CHECK(threw == (seq_steps < 0));
            if (rounds <= 2) CHECK(!threw);
            CHECK(model.missing_pages == 0);
            if (!threw) CHECK(w.items.get_finished_items().size() == 1 && w.items.get_finished_items().front().second.size() == 64);
            std::printf("%s EOF as the 64th token of a row in a pool of 4 pages, %d round(s): sequential %s, pipelined %s\n",
                        threw == (seq_steps < 0) ? "[ OK ]" : "[FAIL]", rounds, seq_steps < 0 ? "stuck" : "finished",
                        threw ? "reported" : "finished");
        }
    }
    {   // a pool that cannot hold even one row: an error, not an endless loop
        World w(4, 64, DEFAULT_INIT_NUM_BLOCKS - 1);
        w.items.add_new_item(IdTokensPair(0, std::vector<int>{1, 2, 3}));
        FakeModel model{4, 64, 0, std::vector<uint64_t>(4, 0), &w.pages};
        bool threw = false;
        try {
            run_paged_engine_pipelined(w.items, w.processing, w.pool, w.pages, 4, 64,
                                       [&](const TensorInt& inp, TensorInt& len, const TensorInt& idx, TensorInt& res, int n_new) {
                                           model.forward(inp, len, idx, res, n_new);
                                       });
        } catch (const std::runtime_error&) {
            threw = true;
        }
        CHECK(threw);
        std::printf("%s pool too small for any row: reported\n", threw ? "[ OK ]" : "[FAIL]");
    }
    // A pool that admits a row (>= DEFAULT_INIT_NUM_BLOCKS pages) but cannot hold its growth, and no EOF: the row is
    // the only one in flight when it runs out of pages.  ADVICE r1: the pipelined loop preempted it, dropped its
    // in-flight token, re-admitted it at the same length and repeated that forever (> 5000 forwards, no progress);
    // the sequential loop ends in "pool too small".  Both must report it, after a bounded number of forwards.
    for (int n_blocks = DEFAULT_INIT_NUM_BLOCKS; n_blocks < 160 / PAGE_BLOCK_SIZE; ++n_blocks) {
        World w(2, 160, n_blocks);
        w.items.add_new_item(IdTokensPair(0, std::vector<int>{1, 2, 3}));
        FakeModel model{2, 160, 0, std::vector<uint64_t>(2, 0), &w.pages};
        bool threw = false;
        try {
            run_paged_engine_pipelined(w.items, w.processing, w.pool, w.pages, 2, 160,
                                       [&](const TensorInt& inp, TensorInt& len, const TensorInt& idx, TensorInt& res, int n_new) {
                                           if (model.launches > 1000) throw std::logic_error("no progress");
                                           model.forward(inp, len, idx, res, n_new);
                                       });
        } catch (const std::runtime_error&) {
            threw = true;
        } catch (const std::logic_error&) {
        }
        CHECK(threw);
        CHECK(model.launches <= n_blocks * PAGE_BLOCK_SIZE + 2);
        std::printf("%s pool of %d pages for a row that needs %d: reported after %lld forwards\n", threw ? "[ OK ]" : "[FAIL]",
                    n_blocks, 160 / PAGE_BLOCK_SIZE, model.launches);
    }
    std::printf("%d failure(s)\n", g_failures);
    return g_failures != 0;
}
