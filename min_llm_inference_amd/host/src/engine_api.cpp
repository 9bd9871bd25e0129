// extern "C" engine sessions (include/mli_engine.h): the reference's engine loops in resumable form, so a
// host in another language -- or bench.py -- can step them and interleave the multi-GPU token gather.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "bf16_extension.h"
#include "constants.h"
#include "fp8_extension.h"
#include "inference_model.h"
#include "inferencer.h"
#include "mli_engine.h"
#include "pipelined_engine.h"
#include "runtime.h"
#include "throughput_counter.h"

namespace {

thread_local std::string g_last_error;
// what an engine created from now on starts with (mli_engine_set_lean_layers / mli_engine_set_step_graphs); every engine
// keeps its own copy, installed for the calling thread while one of its entry points runs
std::atomic<bool> g_default_lean_layers{true};
std::atomic<bool> g_default_step_graphs{false};

TensorFloat upload(const float* host, std::vector<size_t> shape) {
    TensorFloat staging(shape, DeviceType::HOST);
    std::memcpy(staging.data(), host, staging.get_total_size() * sizeof(float));
    TensorFloat device(shape, DeviceType::DEVICE);
    device.copy_from(staging);
    return device;
}

}  // namespace

struct mli_engine {
    mli_engine_config cfg;
    TensorFloat emb_table, pos_table;
    ItemStorage item_storage;
    ProcessingStorage processing_storage;
    std::unique_ptr<InferenceModel> naive_model;
    std::unique_ptr<PagedAttentionInferenceModel> paged_model;
    std::unique_ptr<PagedAttentionCublasInferenceModel> gemm_model;
    std::unique_ptr<PagedAttentionBf16InferenceModel> bf16_model;
    std::unique_ptr<PagedAttentionFp8InferenceModel> fp8_model;
    std::unique_ptr<MemoryBlockManager> pool;
    std::unique_ptr<PagedAttentionsManager> pages;
    TensorInt inp_device, inp_host, lengths_device, lengths_host, new_idx_device, new_idx_host;
    TensorInt result_device, result_host;
    bool started = false;
    int n_new_items = 0;
    long long iterations = 0;
    GemmHandle handle;
    ThroughputCounter counter;  // this engine's own (the reference has one per process)
    int pipelined = -1;         // mli_engine_set_pipelined: 1 / 0 = run() uses the pipelined / the sequential loop,
                                // -1 (default) = pipelined wherever it applies (see use_pipelined())
    // the pipelined loop serves the paged kinds with up to PAGE_BLOCK_SIZE / 2 rounds, without the reference's quirk,
    // and only an engine that is run to completion from the start (mli_engine_step drives the sequential loop)
    bool pipelined_applies() const {
        return cfg.kind != MLI_ENGINE_CONTIGUOUS && 2 * cfg.n_forward_rounds <= PAGE_BLOCK_SIZE &&
               !cfg.reference_length_reset_quirk && !started;
    }
    bool use_pipelined() const { return pipelined == 1 || (pipelined == -1 && pipelined_applies()); }
    void* stream = nullptr;     // private compute stream (mli_engine_use_private_stream), else the thread's
    bool lean_layers = g_default_lean_layers.load();   // this engine's composition and replay switches (runtime.h)
    bool step_graphs = g_default_step_graphs.load();

    ~mli_engine() {
        if (stream) {
            mli::runtime::release_attention_scratch(stream);
            mli::runtime::destroy_stream(stream);
        }
    }

    // every entry point runs under this: device, stream and counter of THIS engine for the calling thread
    struct Scope {
        void* saved;
        bool saved_lean, saved_graphs, saved_sequential;
        explicit Scope(mli_engine* e)
            : saved(mli::runtime::compute_stream()), saved_lean(mli::runtime::lean_layers()),
              saved_graphs(mli::runtime::step_graphs()), saved_sequential(mli::runtime::sequential_engine_loop()) {
            mli::runtime::use_device(e->cfg.device);
            if (e->stream) mli::runtime::set_compute_stream(e->stream);
            mli::runtime::set_lean_layers(e->lean_layers);
            mli::runtime::set_step_graphs(e->step_graphs);
            set_thread_throughput_counter(&e->counter);
        }
        ~Scope() {
            mli::runtime::set_compute_stream(saved);
            mli::runtime::set_lean_layers(saved_lean);
            mli::runtime::set_step_graphs(saved_graphs);
            mli::runtime::set_sequential_engine_loop(saved_sequential);
            set_thread_throughput_counter(nullptr);
        }
    };

    mli_engine(const mli_engine_config& c, const float* emb, const float* pos, const float* wk, const float* wq,
               const float* wv)
        : cfg(c),
          emb_table(upload(emb, {(size_t)c.n_vocab, (size_t)c.emb_dim})),
          pos_table(upload(pos, {(size_t)c.n_sequence, (size_t)c.emb_dim})),
          inp_device({(size_t)c.n_batch, (size_t)c.n_sequence}, DeviceType::DEVICE),
          inp_host({(size_t)c.n_batch, (size_t)c.n_sequence}, DeviceType::HOST),
          lengths_device({(size_t)c.n_batch}, DeviceType::DEVICE), lengths_host({(size_t)c.n_batch}, DeviceType::HOST),
          new_idx_device({(size_t)c.n_batch}, DeviceType::DEVICE), new_idx_host({(size_t)c.n_batch}, DeviceType::HOST),
          result_device(result_shape(c), DeviceType::DEVICE), result_host(result_shape(c), DeviceType::HOST) {
        const size_t B = c.n_batch, S = c.n_sequence, D = c.emb_dim, V = c.n_vocab;
        if (c.kind == MLI_ENGINE_PAGED_BF16) {
            pool = std::make_unique<MemoryBlockManager>(c.n_blocks, bf16_page_block_floats(D));
            pages = std::make_unique<PagedAttentionsManager>(B, S, D);
            bf16_model = std::make_unique<PagedAttentionBf16InferenceModel>(
                PagedAttentionBf16Layer(make_device_bf16(wk, {D, D}), make_device_bf16(wq, {D, D}),
                                        make_device_bf16(wv, {D, D}), B, D, S),
                B, S, D, V, c.n_forward_rounds);
            init_loop_tensors(B, S);
            return;
        }
        if (c.kind == MLI_ENGINE_PAGED_FP8) {
            pool = std::make_unique<MemoryBlockManager>(c.n_blocks, fp8_page_block_floats(D));
            pages = std::make_unique<PagedAttentionsManager>(B, S, D);
            fp8_model = std::make_unique<PagedAttentionFp8InferenceModel>(
                PagedAttentionFp8Layer(make_device_bf16(wk, {D, D}), make_device_bf16(wq, {D, D}),
                                       make_device_bf16(wv, {D, D}), B, D, S),
                B, S, D, V, c.n_forward_rounds);
            init_loop_tensors(B, S);
            return;
        }
        TensorFloat dk = upload(wk, {D, D}), dq = upload(wq, {D, D}), dv = upload(wv, {D, D});
        if (c.kind == MLI_ENGINE_CONTIGUOUS) {
            naive_model = std::make_unique<InferenceModel>(
                SelfAttentionLayer(std::move(dk), std::move(dq), std::move(dv), B, D, S), EncoderLayer(),
                DecoderLayer(B, V), B, S, D);
        } else {
            pool = std::make_unique<MemoryBlockManager>(c.n_blocks, (size_t)PAGE_BLOCK_SIZE * 3 * D);
            pages = std::make_unique<PagedAttentionsManager>(B, S, D);
            if (c.kind == MLI_ENGINE_PAGED)
                paged_model = std::make_unique<PagedAttentionInferenceModel>(
                    PagedAttentionLayer(std::move(dk), std::move(dq), std::move(dv), B, D, S), PagedEncoderLayer(),
                    PagedDecoderLayer(B, V), B, S, D, c.n_forward_rounds);
            else
                gemm_model = std::make_unique<PagedAttentionCublasInferenceModel>(
                    PagedAttentionCublasLayer(std::move(dk), std::move(dq), std::move(dv), B, D, S),
                    PagedEncoderLayer(), PagedCublasDecoderLayer(B, V), B, S, D, c.n_forward_rounds);
        }
        init_loop_tensors(B, S);
    }

    void init_loop_tensors(size_t B, size_t S) {
        std::memset(lengths_host.data(), 0, B * sizeof(int));
        std::memset(inp_host.data(), 0, B * S * sizeof(int));
        inp_device.copy_from(inp_host);
        lengths_device.copy_from(lengths_host);
    }

    static std::vector<size_t> result_shape(const mli_engine_config& c) {
        if (c.kind == MLI_ENGINE_CONTIGUOUS) return {(size_t)c.n_batch};
        return {(size_t)c.n_batch, (size_t)c.n_forward_rounds};
    }

    bool paged() const { return cfg.kind != MLI_ENGINE_CONTIGUOUS; }

    void insert(const std::vector<int>& free_slots) {
        if (paged()) {
            n_new_items = (int)insert_new_items(inp_device, inp_host, lengths_device, lengths_host, new_idx_device,
                                                new_idx_host, item_storage, processing_storage, *pool, *pages,
                                                cfg.n_forward_rounds).size();
        } else {
            n_new_items = insert_new_items(free_slots, inp_device, inp_host, lengths_device, lengths_host,
                                           new_idx_device, new_idx_host, item_storage, processing_storage);
        }
    }

    void start() {
        if (pages) pages->set_length_reset_quirk(cfg.reference_length_reset_quirk != 0);
        get_global_throughput_counter().reset();
        get_global_throughput_counter().start_record();
        std::vector<int> all(cfg.n_batch);
        std::iota(all.begin(), all.end(), 0);
        insert(all);
        started = true;
    }

    bool done() { return is_done(item_storage, processing_storage); }

    void run_pipelined() {
        if (started) throw std::runtime_error("pipelined run on an engine that has already been stepped");
        if (!paged() || 2 * cfg.n_forward_rounds > PAGE_BLOCK_SIZE)
            throw std::runtime_error("the pipelined loop serves the paged kinds with n_forward_rounds <= PAGE_BLOCK_SIZE / 2");
        pages->set_length_reset_quirk(cfg.reference_length_reset_quirk != 0);
        get_global_throughput_counter().reset();
        started = true;
        iterations = run_paged_engine_pipelined(
            item_storage, processing_storage, *pool, *pages, cfg.n_batch, cfg.n_sequence,
            [&](const TensorInt& inp, TensorInt& lengths, const TensorInt& new_idx, TensorInt& result, int n_new) {
                if (cfg.kind == MLI_ENGINE_PAGED)
                    paged_model->forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                         pages->get_page_table_device());
                else if (cfg.kind == MLI_ENGINE_PAGED_BF16)
                    bf16_model->forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                        pages->get_page_table_device());
                else if (cfg.kind == MLI_ENGINE_PAGED_FP8)
                    fp8_model->forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                       pages->get_page_table_device());
                else
                    gemm_model->forward(inp, lengths, new_idx, result, n_new, emb_table, pos_table,
                                        pages->get_page_table_device(), handle);
            }, cfg.n_forward_rounds);
    }

    double t_forward = 0, t_result = 0, t_pages = 0, t_insert = 0;  // host seconds per phase (MLI_ENGINE_TIMING=1)
    static double now() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    void step() {
        if (!started) start();
        if (done()) return;
        const double t0 = now();
        if (cfg.kind == MLI_ENGINE_CONTIGUOUS)
            naive_model->forward(inp_device, lengths_device, new_idx_device, result_device, n_new_items, emb_table,
                                 pos_table);
        else if (cfg.kind == MLI_ENGINE_PAGED)
            paged_model->forward(inp_device, lengths_device, new_idx_device, result_device, n_new_items, emb_table,
                                 pos_table, pages->get_page_table_device());
        else if (cfg.kind == MLI_ENGINE_PAGED_BF16)
            bf16_model->forward(inp_device, lengths_device, new_idx_device, result_device, n_new_items, emb_table,
                                pos_table, pages->get_page_table_device());
        else if (cfg.kind == MLI_ENGINE_PAGED_FP8)
            fp8_model->forward(inp_device, lengths_device, new_idx_device, result_device, n_new_items, emb_table,
                               pos_table, pages->get_page_table_device());
        else
            gemm_model->forward(inp_device, lengths_device, new_idx_device, result_device, n_new_items, emb_table,
                                pos_table, pages->get_page_table_device(), handle);
        const double t1 = now();
        std::vector<int> free_slots =
            process_decoder_result(result_device, result_host, item_storage, processing_storage, cfg.n_sequence);
        const double t2 = now();
        if (paged())
            allocate_or_free_memory_blocks_if_needed(*pages, *pool, processing_storage, item_storage, free_slots,
                                                     cfg.n_forward_rounds);
        const double t3 = now();
        insert(free_slots);
        if (paged() && processing_storage.size() == 0 && item_storage.new_count() > 0)
            throw std::runtime_error("paged engine: the page pool is too small for the next queued item");
        const double t4 = now();
        t_forward += t1 - t0; t_result += t2 - t1; t_pages += t3 - t2; t_insert += t4 - t3;
        ++iterations;
    }

    void fill(mli_engine_stats* s) {
        s->total_tokens = get_global_throughput_counter().total_tokens();
        s->seconds = get_global_throughput_counter().seconds();
        s->iterations = iterations;
        s->finished = item_storage.finish_count();
        s->waiting = item_storage.new_count();
        s->in_flight = processing_storage.size();
    }
};

#define MLI_GUARD(body)                                  \
    try {                                                \
        body;                                            \
        return 0;                                        \
    } catch (const std::exception& e) {                  \
        g_last_error = e.what();                         \
        return -1;                                       \
    } catch (...) {                                      \
        g_last_error = "unknown C++ exception";          \
        return -1;                                       \
    }

extern "C" {

void mli_engine_set_lean_layers(int enabled) { g_default_lean_layers.store(enabled != 0); }

void mli_engine_set_step_graphs(int enabled) { g_default_step_graphs.store(enabled != 0); }

int mli_engine_configure(mli_engine* e, int lean_layers, int step_graphs) {
    MLI_GUARD({
        if (e->started) throw std::runtime_error("mli_engine_configure after the engine has started");
        if (lean_layers >= 0) e->lean_layers = lean_layers != 0;
        if (step_graphs >= 0) e->step_graphs = step_graphs != 0;
        if (e->cfg.kind == MLI_ENGINE_PAGED_FP8 && !e->lean_layers)
            throw std::runtime_error("the fp8 engine has the lean compositions only");
    })
}

const char* mli_engine_last_error(void) { return g_last_error.c_str(); }

int mli_engine_create(const mli_engine_config* c, const float* emb_table, const float* pos_table, const float* wk,
                      const float* wq, const float* wv, mli_engine** out) {
    if (!c || !out || !emb_table || !pos_table || !wk || !wq || !wv) { g_last_error = "null argument"; return -1; }
    if (c->kind < 0 || c->kind > MLI_ENGINE_PAGED_FP8 || (c->kind == MLI_ENGINE_PAGED_BF16 && c->emb_dim % 8) ||
        (c->kind == MLI_ENGINE_PAGED_FP8 && (c->emb_dim % 16 || c->emb_dim > 2048)) || c->n_batch <= 0 || c->n_sequence <= 0 || c->emb_dim <= 0 || c->emb_dim % 4 ||
        c->n_vocab <= EOF_TOKEN_ID ||
        (c->kind != MLI_ENGINE_CONTIGUOUS && (c->n_sequence % PAGE_BLOCK_SIZE || c->n_blocks <= 0 ||
                                              c->n_forward_rounds < 1 || c->n_forward_rounds > PAGE_BLOCK_SIZE)) ||
        (c->kind == MLI_ENGINE_CONTIGUOUS && c->n_sequence % 4)) {
        g_last_error = "invalid engine configuration";
        return -1;
    }
    MLI_GUARD({
        mli::runtime::use_device(c->device);
        *out = new mli_engine(*c, emb_table, pos_table, wk, wq, wv);
    })
}

void mli_engine_destroy(mli_engine* e) {
    if (e && std::getenv("MLI_ENGINE_TIMING") && e->iterations && e->t_forward > 0)
        std::fprintf(stderr, "[mli engine] %lld iterations; host us/iteration: launch forward %.1f, wait + process "
                     "decoder result %.1f, page bookkeeping %.1f, insert + uploads %.1f\n", e->iterations,
                     1e6 * e->t_forward / e->iterations, 1e6 * e->t_result / e->iterations,
                     1e6 * e->t_pages / e->iterations, 1e6 * e->t_insert / e->iterations);
    delete e;
}

int mli_engine_add_item(mli_engine* e, int id, const int* tokens, int n_tokens) {
    if (!e || !tokens || n_tokens <= 0 || n_tokens + 1 > e->cfg.n_sequence) { g_last_error = "bad item"; return -1; }
    MLI_GUARD(e->item_storage.add_new_item(std::make_pair(id, std::vector<int>(tokens, tokens + n_tokens))))
}

int mli_engine_use_private_stream(mli_engine* e) {
    MLI_GUARD({
        mli::runtime::use_device(e->cfg.device);
        if (!e->stream) e->stream = mli::runtime::create_stream();
    })
}

int mli_engine_set_pipelined(mli_engine* e, int enabled) {
    MLI_GUARD({
        if (e->started) throw std::runtime_error("set_pipelined after the engine has started");
        e->pipelined = enabled != 0 ? 1 : 0;
    })
}

int mli_engine_step(mli_engine* e, int* done) {
    MLI_GUARD({
        if (e->pipelined == 1) throw std::runtime_error("a pipelined engine is run to completion (mli_engine_run)");
        mli_engine::Scope scope(e);
        e->step();
        if (done) *done = e->done() ? 1 : 0;
    })
}

int mli_engine_run(mli_engine* e, mli_engine_stats* stats) {
    MLI_GUARD({
        mli_engine::Scope scope(e);
        if (e->use_pipelined()) {
            e->run_pipelined();
        } else {
            if (!e->started) e->start();
            while (!e->done()) e->step();
        }
        if (stats) e->fill(stats);
    })
}

int mli_engine_get_stats(mli_engine* e, mli_engine_stats* stats) {
    MLI_GUARD({
        mli_engine::Scope scope(e);
        e->fill(stats);
    })
}

int mli_engine_decoder_result(mli_engine* e, void** device_ptr, int* count) {
    MLI_GUARD({
        *device_ptr = e->result_device.data();
        *count = (int)e->result_device.get_total_size();
    })
}

int mli_engine_stream(mli_engine* e, void** stream) {
    MLI_GUARD({
        if (!e || !stream) throw std::runtime_error("null argument");
        *stream = e->stream;
    })
}

int mli_engine_get_finished(mli_engine* e, int index, int* id, int* tokens, int capacity, int* n_tokens) {
    MLI_GUARD({
        const auto& items = e->item_storage.get_finished_items();
        if (index < 0 || index >= (int)items.size()) throw std::out_of_range("finished item index");
        auto it = std::next(items.begin(), index);
        *id = it->first;
        *n_tokens = (int)it->second.size();
        if (tokens) std::memcpy(tokens, it->second.data(), sizeof(int) * std::min<size_t>(capacity, it->second.size()));
    })
}

}  // extern "C"
