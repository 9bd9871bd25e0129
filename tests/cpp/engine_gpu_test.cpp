// C++ drop-in check on a real GPU: a driver written against the reference's own API (the shape of
// tests/inferencer_test.cpp:12-125 and tests/paged_for_profile.cpp:10-62) compiled with g++ against
// min_llm_inference_amd/host/include and linked with libmli_hip.so -- no HIP header, no Python.
// Pass criterion = the reference's: every item finishes (finish_count == n_items); plus contiguous, paged and
// paged-"cublas" engines must agree token for token.
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "constants.h"
#include "inference_model.h"
#include "inferencer.h"
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
    int mismatched = 0;
    for (const auto& kv : naive)
        if (paged[kv.first] != kv.second || gemm[kv.first] != kv.second) ++mismatched;
    std::printf("items whose tokens differ between engines: %d\n", mismatched);
    failures += mismatched != 0;
    std::printf("%s\n", failures ? "FAILED" : "ALL ENGINES AGREE");
    return failures ? 1 : 0;
}
