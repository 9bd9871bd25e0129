// C++ drop-in check on a real GPU: a driver written against the reference's own API (the shape of
// tests/inferencer_test.cpp:12-125 and tests/paged_for_profile.cpp:10-62) compiled with g++ against
// min_llm_inference_amd/host/include and linked with libmli_hip.so -- no HIP header, no Python.
// Pass criterion = the reference's: every item finishes (finish_count == n_items); plus contiguous, paged and
// paged-"cublas" engines must agree token for token.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "constants.h"
#include "inference_model.h"
#include "inferencer.h"
#include "pipelined_engine.h"
#include "runtime.h"
#include "throughput_counter.h"

static std::mt19937 rng(4711);

static TensorFloat random_device_tensor(std::vector<size_t> shape, float scale) {
    TensorFloat host(shape, DeviceType::HOST);
    std::uniform_real_distribution<float> u(-scale, scale);
    for (size_t i = 0; i < host.get_total_size(); ++i) host.data()[i] = u(rng);
    TensorFloat dev(shape, DeviceType::DEVICE);
    dev.copy_from(host);
    return dev;
}

static TensorFloat clone(const TensorFloat& t) {
    TensorFloat c(t.shape(), DeviceType::DEVICE);
    c.copy_from(t);
    return c;
}

static std::map<int, std::vector<int>> collect(const ItemStorage& s) {
    std::map<int, std::vector<int>> out;
    for (const auto& it : s.get_finished_items()) out[it.first] = it.second;
    return out;
}

int main() {
    const size_t B = 96, S = 160, D = 256, V = 1100;
    const int n_items = 2 * B + 5;
    std::vector<IdTokensPair> items;
    for (int i = 0; i < n_items; ++i) {
        std::vector<int> toks(1 + rng() % 63);
        for (int& t : toks) t = rng() % EOF_TOKEN_ID;
        items.emplace_back(i, toks);
    }
    TensorFloat emb = random_device_tensor({V, D}, 1.0f), pos = random_device_tensor({S, D}, 0.5f);
    const float ws = 2.0f / 16.0f;
    TensorFloat wk = random_device_tensor({D, D}, ws), wq = random_device_tensor({D, D}, ws), wv = random_device_tensor({D, D}, ws);
    int failures = 0;

    // 1. contiguous engine (start_inference_engine)
    std::map<int, std::vector<int>> naive;
    {
        ItemStorage storage;
        ProcessingStorage processing;
        for (const auto& it : items) storage.add_new_item(IdTokensPair(it));
        InferenceModel model(SelfAttentionLayer(clone(wk), clone(wq), clone(wv), B, D, S), EncoderLayer(), DecoderLayer(B, V), B, S, D);
        start_inference_engine(emb, pos, storage, processing, model, B, S);
        std::printf("contiguous engine: finished %d of %d\n", storage.finish_count(), n_items);
        failures += storage.finish_count() != n_items;
        naive = collect(storage);
    }
    // 2. paged engine, 4 pages per slot (growth + preemption), as tests/inferencer_test.cpp:46-74
    std::map<int, std::vector<int>> paged;
    {
        ItemStorage storage;
        ProcessingStorage processing;
        for (const auto& it : items) storage.add_new_item(IdTokensPair(it));
        MemoryBlockManager pool(DEFAULT_INIT_NUM_BLOCKS * B, PAGE_BLOCK_SIZE * 3 * D);
        PagedAttentionsManager pages(B, S, D);
        PagedAttentionInferenceModel model(PagedAttentionLayer(clone(wk), clone(wq), clone(wv), B, D, S), PagedEncoderLayer(),
                                           PagedDecoderLayer(B, V), B, S, D, 1);
        start_paged_attention_inference_engine(emb, pos, storage, processing, pool, pages, model, B, S, 1);
        std::printf("paged engine: finished %d of %d\n", storage.finish_count(), n_items);
        failures += storage.finish_count() != n_items;
        paged = collect(storage);
    }
    // 2b. the same engine with the layers on the reference's launch sequence (scores, probabilities and emb_score
    //     materialised) instead of the lean compositions: the tokens must not notice
    std::map<int, std::vector<int>> paged_reference_sequence;
    {
        mli::runtime::set_lean_layers(false);
        mli::runtime::set_sequential_engine_loop(true);   // ... and in the reference's loop order
        ItemStorage storage;
        ProcessingStorage processing;
        for (const auto& it : items) storage.add_new_item(IdTokensPair(it));
        MemoryBlockManager pool(DEFAULT_INIT_NUM_BLOCKS * B, PAGE_BLOCK_SIZE * 3 * D);
        PagedAttentionsManager pages(B, S, D);
        PagedAttentionInferenceModel model(PagedAttentionLayer(clone(wk), clone(wq), clone(wv), B, D, S), PagedEncoderLayer(),
                                           PagedDecoderLayer(B, V), B, S, D, 1);
        start_paged_attention_inference_engine(emb, pos, storage, processing, pool, pages, model, B, S, 1);
        mli::runtime::set_lean_layers(true);
        mli::runtime::set_sequential_engine_loop(false);
        std::printf("paged engine, sequential loop, materialising layers: finished %d of %d\n", storage.finish_count(), n_items);
        failures += storage.finish_count() != n_items;
        paged_reference_sequence = collect(storage);
    }
    // 3. paged "cublas" engine (GemmHandle in place of cublasHandle_t), roomy pool, 2 forward rounds
    std::map<int, std::vector<int>> gemm;
    {
        ItemStorage storage;
        ProcessingStorage processing;
        for (const auto& it : items) storage.add_new_item(IdTokensPair(it));
        MemoryBlockManager pool((S / PAGE_BLOCK_SIZE) * B, PAGE_BLOCK_SIZE * 3 * D);
        PagedAttentionsManager pages(B, S, D);
        PagedAttentionCublasInferenceModel model(PagedAttentionCublasLayer(clone(wk), clone(wq), clone(wv), B, D, S),
                                                 PagedEncoderLayer(), PagedCublasDecoderLayer(B, V), B, S, D, 2);
        start_paged_attention_cublas_inference_engine(emb, pos, storage, processing, pool, pages, model, B, S, 2);
        std::printf("paged gemm engine (2 rounds): finished %d of %d\n", storage.finish_count(), n_items);
        failures += storage.finish_count() != n_items;
        gemm = collect(storage);
    }
    // 4. pipelined loop (extension), tight pool: preemption while tokens are in flight
    std::map<int, std::vector<int>> pipelined;
    {
        ItemStorage storage;
        ProcessingStorage processing;
        for (const auto& it : items) storage.add_new_item(IdTokensPair(it));
        MemoryBlockManager pool(DEFAULT_INIT_NUM_BLOCKS * B, PAGE_BLOCK_SIZE * 3 * D);
        PagedAttentionsManager pages(B, S, D);
        PagedAttentionCublasInferenceModel model(PagedAttentionCublasLayer(clone(wk), clone(wq), clone(wv), B, D, S),
                                                 PagedEncoderLayer(), PagedCublasDecoderLayer(B, V), B, S, D, 1);
        start_paged_attention_cublas_inference_engine_pipelined(emb, pos, storage, processing, pool, pages, model, B, S);
        std::printf("pipelined paged gemm engine: finished %d of %d\n", storage.finish_count(), n_items);
        failures += storage.finish_count() != n_items;
        pipelined = collect(storage);
    }
    // 5. bf16 pages (extension): finishes, prompts intact (tokens differ from fp32 by design)
    {
        ItemStorage storage;
        ProcessingStorage processing;
        for (const auto& it : items) storage.add_new_item(IdTokensPair(it));
        MemoryBlockManager pool(DEFAULT_INIT_NUM_BLOCKS * B, bf16_page_block_floats(D));
        PagedAttentionsManager pages(B, S, D);
        TensorFloat hk({D, D}, DeviceType::HOST), hq({D, D}, DeviceType::HOST), hv({D, D}, DeviceType::HOST);
        hk.copy_from(wk); hq.copy_from(wq); hv.copy_from(wv);
        PagedAttentionBf16InferenceModel model(
            PagedAttentionBf16Layer(make_device_bf16(hk.data(), {D, D}), make_device_bf16(hq.data(), {D, D}),
                                    make_device_bf16(hv.data(), {D, D}), B, D, S), B, S, D, V, 1);
        start_paged_attention_bf16_inference_engine(emb, pos, storage, processing, pool, pages, model, B, S, 1);
        std::printf("bf16 paged engine: finished %d of %d\n", storage.finish_count(), n_items);
        failures += storage.finish_count() != n_items;
        int broken = 0;
        for (const auto& kv : collect(storage)) {
            const std::vector<int>& prompt = items[kv.first].second;
            if (kv.second.size() <= prompt.size() || !std::equal(prompt.begin(), prompt.end(), kv.second.begin())) ++broken;
        }
        std::printf("bf16 items with a damaged prompt: %d\n", broken);
        failures += broken != 0;
    }
    // 6. the async-allocation Tensor flavour (reference tensor.hpp:182-269): allocate, copy H2D and back on the
    //    transfer stream, data() waits for readiness
    {
        Tensor<float> h({1 << 16}, DeviceType::HOST, TensorDataType::ASYNC_ALLOCATE), back({1 << 16}, DeviceType::HOST, TensorDataType::ASYNC_ALLOCATE);
        Tensor<float> d({1 << 16}, DeviceType::DEVICE, TensorDataType::ASYNC_ALLOCATE);
        for (size_t i = 0; i < h.get_total_size(); ++i) h.data()[i] = static_cast<float>(i) * 0.5f;
        d.copy_from(h);
        back.copy_from(d);
        int bad = 0;
        for (size_t i = 0; i < back.get_total_size(); ++i) bad += back.data()[i] != static_cast<float>(i) * 0.5f;
        std::printf("async-allocation tensor round trip: %d mismatches\n", bad);
        failures += bad != 0;
    }
    // 7. start_paged_* with a pool that cannot serve the queue must throw, never launch a forward over undefined
    //    lengths / page pointers and never spin: (a) the first insert admits nothing -- the pool is
    //    below DEFAULT_INIT_NUM_BLOCKS; (b) the row is admitted but the pool cannot hold its growth, sequential and
    //    pipelined loop alike.
    for (int scenario = 0; scenario < 3; ++scenario) {
        ItemStorage storage;
        ProcessingStorage processing;
        storage.add_new_item(IdTokensPair(0, std::vector<int>{1, 2, 3}));
        const int n_blocks = scenario == 0 ? DEFAULT_INIT_NUM_BLOCKS - 1 : DEFAULT_INIT_NUM_BLOCKS + 1;
        MemoryBlockManager pool(n_blocks, PAGE_BLOCK_SIZE * 3 * D);
        PagedAttentionsManager pages(2, S, D);
        // an embedding table whose EOF row is zero: greedy decoding never ends a row early, so it must outgrow the pool
        TensorFloat emb_host({V, D}, DeviceType::HOST), emb_no_eof({V, D}, DeviceType::DEVICE);
        emb_host.copy_from(emb);
        std::memset(emb_host.data() + static_cast<size_t>(EOF_TOKEN_ID) * D, 0, D * sizeof(float));
        emb_no_eof.copy_from(emb_host);
        PagedAttentionInferenceModel model(PagedAttentionLayer(clone(wk), clone(wq), clone(wv), 2, D, S), PagedEncoderLayer(),
                                           PagedDecoderLayer(2, V), 2, S, D, 1);
        bool threw = false;
        mli::runtime::set_sequential_engine_loop(scenario != 2);
        try {
            if (scenario == 2)
                start_paged_attention_inference_engine_pipelined(emb_no_eof, pos, storage, processing, pool, pages, model, 2, S);
            else
                start_paged_attention_inference_engine(emb_no_eof, pos, storage, processing, pool, pages, model, 2, S, 1);
        } catch (const std::runtime_error& e) {
            threw = std::strstr(e.what(), "too small") != nullptr;
        }
        mli::runtime::set_sequential_engine_loop(false);
        std::printf("pool of %d pages, %s loop: %s\n", n_blocks, scenario == 2 ? "pipelined" : "sequential",
                    threw ? "reported as too small" : "NOT reported");
        failures += !threw;
    }
    int mismatched = 0;
    for (const auto& kv : naive)
        if (paged[kv.first] != kv.second || gemm[kv.first] != kv.second || pipelined[kv.first] != kv.second ||
            paged_reference_sequence[kv.first] != kv.second)
            ++mismatched;
    std::printf("items whose tokens differ between engines: %d\n", mismatched);
    failures += mismatched != 0;
    // one number over every item's tokens: the two allocation flavours (this file built with and without
    // -DDEFAULT_ALLOC_METHOD=1, as the reference's Makefile:20-30 runs its suite) must print the same
    unsigned long long checksum = 1469598103934665603ull;
    for (const auto& kv : naive) {
        checksum = (checksum ^ (unsigned long long)kv.first) * 1099511628211ull;
        for (int t : kv.second) checksum = (checksum ^ (unsigned long long)(unsigned)t) * 1099511628211ull;
    }
    std::printf("allocation flavour: %s (DEFAULT_ALLOC_METHOD=%d); process default seen by the library: %s\n",
                DEFAULT_ALLOC_METHOD ? "ASYNC_ALLOCATE" : "SYNC_ALLOCATE", DEFAULT_ALLOC_METHOD,
                mli::mem::process_default_mode() == mli::mem::Mode::Async ? "async" : "sync");
    failures += (mli::mem::process_default_mode() == mli::mem::Mode::Async) != (DEFAULT_ALLOC_METHOD != 0);
    std::printf("TOKENS CHECKSUM %016llx\n", checksum);
    std::printf("%s\n", failures ? "FAILED" : "ALL ENGINES AGREE");
    return failures ? 1 : 0;
}
