/*
 * mli_engine.h -- C ABI over the host engine loops (continuous batching + paged KV allocator) of
 * libmli_hip.so.  It binds what the reference exposes as C++ only:
 *   start_inference_engine / start_paged_attention_inference_engine /
 *   start_paged_attention_cublas_inference_engine       (reference include/inferencer.h:18-32)
 * together with the objects their callers build first (ItemStorage, ProcessingStorage,
 * MemoryBlockManager, PagedAttentionsManager, *InferenceModel; reference tests/paged_for_profile.cpp:10-62).
 * One engine drives one GPU.  bench.py and the tests use it through ctypes; a C++ host links the classes
 * directly (min_llm_inference_amd/host/include).
 */
#ifndef MLI_ENGINE_H
#define MLI_ENGINE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mli_engine mli_engine;

/* MLI_ENGINE_PAGED_BF16 is an EXTENSION (the reference is fp32 only): the paged GEMM engine over bf16 pages and
 * bf16 Wk/Wq/Wv (the fp32 host weights are rounded to nearest-even at creation); emb_dim % 8 == 0; a page is
 * 16 * 3 * emb_dim bf16 elements. */
/* MLI_ENGINE_PAGED_FP8 is an EXTENSION, opt-in: OCP e4m3 pages (x, K and V: one byte per element under
 * the same layout rule, a page is 16 * 3 * emb_dim bytes), bf16 Wk/Wq/Wv, fp32 everywhere else; emb_dim % 16 == 0; lean
 * compositions only. */
enum { MLI_ENGINE_CONTIGUOUS = 0, MLI_ENGINE_PAGED = 1, MLI_ENGINE_PAGED_GEMM = 2, MLI_ENGINE_PAGED_BF16 = 3,
       MLI_ENGINE_PAGED_FP8 = 4 };

typedef struct {
    int kind;             /* MLI_ENGINE_* */
    int n_batch;          /* slots in the continuous batch */
    int n_sequence;       /* max tokens per item (multiple of 16 for the paged kinds) */
    int emb_dim;          /* multiple of 4 */
    int n_vocab;
    int n_blocks;         /* paged: pages in the pool (each 16 * 3 * emb_dim floats) */
    int n_forward_rounds; /* paged: decode rounds per iteration, 1..16 */
    int device;           /* GPU ordinal this engine runs on */
    int reference_length_reset_quirk; /* 1 = reproduce src/paged_item_storage.cpp:110-118 (measurement only) */
} mli_engine_config;

typedef struct {
    long long total_tokens; /* tokens appended by process_decoder_result (ThroughputCounter) */
    double seconds;         /* wall time since the first insert, host work and copies included */
    long long iterations;   /* engine iterations executed */
    int finished;           /* items finished */
    int waiting;            /* items still queued */
    int in_flight;          /* items occupying a slot */
} mli_engine_stats;

/* Weights are HOST pointers (row-major fp32): emb_table [n_vocab, emb_dim], pos_table [n_sequence, emb_dim],
 * wk/wq/wv [emb_dim, emb_dim]; they are copied to the device.  Returns 0 or a negative error
 * (mli_engine_last_error() has the message). */
int mli_engine_create(const mli_engine_config* config, const float* emb_table, const float* pos_table,
                      const float* wk, const float* wq, const float* wv, mli_engine** out);
void mli_engine_destroy(mli_engine* engine);

/* Give the engine its own non-blocking HIP stream (default: the calling thread's stream, i.e. the legacy default
 * stream, as the reference).  Engines with private streams can be driven from different threads of one process
 * and overlap on one GPU: while one waits for its decoder result and does its host bookkeeping, the other's
 * kernels run.  Each engine has its own ThroughputCounter, scratch and page pool; an engine is driven by one
 * thread at a time. */
int mli_engine_use_private_stream(mli_engine* engine);

/* EXTENSION: the pipelined loop of the paged kinds -- the host works one step behind the GPU (page
 * growth and admission for step k+1 while step k's result is still in flight; per-slot device updates instead of
 * whole-tensor uploads), min_llm_inference_amd/host/include/pipelined_engine.h.  Per-item token streams are identical
 * to the sequential loop's.  It is what mli_engine_run uses BY DEFAULT wherever it applies: a paged kind,
 * n_forward_rounds <= 8, no reference_length_reset_quirk, an engine that has not been stepped.  enabled = 0 selects
 * the reference's sequential loop order, enabled = 1 insists on the pipelined loop (mli_engine_run then fails where it
 * does not apply, and mli_engine_step is refused).  Call before the first run. */
int mli_engine_set_pipelined(mli_engine* engine, int enabled);

/* Queue one item (ItemStorage::add_new_item). */
int mli_engine_add_item(mli_engine* engine, int id, const int* tokens, int n_tokens);

/* Run to completion (the reference's start_*_engine). */
int mli_engine_run(mli_engine* engine, mli_engine_stats* stats);

/* One iteration: forward -> process_decoder_result -> page bookkeeping -> insert_new_items.  The first call
 * also performs the initial insert.  *done is set to 1 once every item has finished. */
int mli_engine_step(mli_engine* engine, int* done);

int mli_engine_get_stats(mli_engine* engine, mli_engine_stats* stats);

/* Device pointer to the int32 decoder output of the last iteration, [n_batch, n_forward_rounds]
 * (what a multi-GPU host all-gathers), and its element count. */
int mli_engine_decoder_result(mli_engine* engine, void** device_ptr, int* count);

/* The stream this engine's kernels run on (hipStream_t as void*; NULL = the legacy default stream): what a multi-GPU host
 * enqueues the token all-gather on (include/mli_shard.h). */
int mli_engine_stream(mli_engine* engine, void** stream);

/* Finished item `index` (0 <= index < stats.finished), in completion order: id and tokens (prompt + generated). */
int mli_engine_get_finished(mli_engine* engine, int index, int* id, int* tokens, int capacity, int* n_tokens);

/* The DEFAULT of engines created afterwards (every engine keeps its own value, see mli_engine_configure; two engines in one
 * process never change each other's composition): 1 (default) = the models' layers run the lean compositions (prefill with the encoder as the fill GEMM's
 * prologue, attention without materialised scores / probabilities, decoder head with the argmax as the logits GEMM's
 * epilogue), 0 = the reference's launch sequence (encoder, fill, latest, scan + combine, logits, argmax).  Tokens are
 * identical either way; the switch exists to measure one against the other. */
void mli_engine_set_lean_layers(int enabled);

/* The DEFAULT of engines created afterwards: 1 = the models replay their pure decode forwards (no newly inserted rows) from a hipGraph recorded on
 * the first such forward -- one host call per forward instead of one per launch (host/include/step_graph.h).  Only
 * engines with a private stream can record (the legacy default stream cannot be captured); others keep launching
 * eagerly.  Default 0: a replay costs the GPU a few microseconds more than the same launches issued from C++. */
void mli_engine_set_step_graphs(int enabled);

/* This engine's own switches, before its first step / run: lean_layers and step_graphs as above; -1 leaves a value as
 * it is.  (The fp8 engine has the lean compositions only.) */
int mli_engine_configure(mli_engine* engine, int lean_layers, int step_graphs);

const char* mli_engine_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MLI_ENGINE_H */
