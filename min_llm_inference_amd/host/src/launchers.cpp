// C++ adapters with the reference's launcher signatures over the C ABI (include/mli_kernels.h).
// They unpack Tensor shapes exactly the way the reference launchers do, pass the calling thread's
// compute stream, and turn a non-zero status into the reference's error behaviour (print + throw).
#include <cassert>
#include <iterator>
#include <map>
#include <memory>
#include <mutex>

#include "constants.h"
#include "kernels/decoder.h"
#include "kernels/encoder.h"
#include "kernels/paged_attention.h"
#include "kernels/self_attention_inference_optimized.h"
#include "mli_kernels.h"
#include "runtime.h"
#include "step_graph.h"
#include "utils.h"

// Split-sequence scratch, one buffer per (device, compute stream) -- kernels of two streams may run at the same
// time --, grown on demand (never shrinks).  It lives on the host
// side because the C ABI itself never allocates.
namespace {
std::mutex g_scratch_mu;
std::map<std::pair<int, void*>, std::unique_ptr<Tensor<char>>> g_scratch;  // (device, stream)
}  // namespace

void mli::runtime::release_attention_scratch(void* stream) noexcept {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto it = g_scratch.begin(); it != g_scratch.end();) it = it->first.second == stream ? g_scratch.erase(it) : std::next(it);
}

mli::runtime::Scratch mli::runtime::attention_scratch(int n_batch, int n_sequence, int dim) {
    std::mutex& mu = g_scratch_mu;
    auto& per_device = g_scratch;
    const size_t need = mli_attention_workspace_bytes(n_batch, n_sequence, dim);
    if (need == 0) return {nullptr, 0};
    std::lock_guard<std::mutex> lock(mu);
    auto& slot = per_device[{mli::runtime::current_device(), mli::runtime::compute_stream()}];
    if (!slot || slot->get_total_size() < need) {
        slot = std::make_unique<Tensor<char>>(std::vector<size_t>{need}, DeviceType::DEVICE,
                                              TensorDataType::SYNC_ALLOCATE);
        // the row arrival counters at the front start at zero (once per allocation; the kernels keep them there)
        HIP_CHECK(mli_attention_workspace_init(slot->data(), need, mli::runtime::compute_stream()));
    }
    return {slot->data(), need};
}

// Per THREAD, like the device and the compute stream (an engine is driven by one thread at a time, and the engine C ABI
// installs each engine's own values for the duration of a call: engine_api.cpp, mli_engine::Scope): two engines in one
// process cannot change each other's composition.
namespace {
thread_local bool g_lean_layers = true;
thread_local bool g_sequential_engine_loop = false;
thread_local bool g_step_graphs = false;
}
void mli::runtime::set_step_graphs(bool enabled) { g_step_graphs = enabled; }
bool mli::runtime::step_graphs() { return g_step_graphs; }

StepGraph::~StepGraph() { reset(); }
StepGraph::StepGraph(StepGraph&& other) noexcept
    : exec_(other.exec_), key_(std::move(other.key_)), seen_(other.seen_), replayed_(false) {
    other.exec_ = nullptr;
    other.seen_ = 0;
}
void StepGraph::reset() noexcept {
    if (exec_) (void)mli_graph_destroy(exec_);
    exec_ = nullptr;
    seen_ = 0;
}
bool StepGraph::begin(const std::vector<const void*>& key) {
    replayed_ = false;
    void* st = mli::runtime::compute_stream();
    if (!mli::runtime::step_graphs() || st == nullptr) return false;
    std::vector<const void*> full(key);
    full.push_back(st);
    if (full != key_) {
        reset();
        key_ = full;
    }
    if (exec_ != nullptr) {
        HIP_CHECK(mli_graph_launch(exec_, st));
        replayed_ = true;
        return false;
    }
    if (seen_++ == 0) return false;   // first forward over these buffers: eager, scratch gets allocated
    HIP_CHECK(mli_graph_begin_capture(st));
    return true;
}
void StepGraph::finish() {
    void* st = mli::runtime::compute_stream();
    HIP_CHECK(mli_graph_end_capture(st, &exec_));
    HIP_CHECK(mli_graph_launch(exec_, st));
}
void StepGraph::abandon() noexcept {
    void* discard = nullptr;
    (void)mli_graph_end_capture(mli::runtime::compute_stream(), &discard);
    if (discard) (void)mli_graph_destroy(discard);
    reset();
}
void mli::runtime::set_sequential_engine_loop(bool enabled) { g_sequential_engine_loop = enabled; }
bool mli::runtime::sequential_engine_loop() { return g_sequential_engine_loop; }
void mli::runtime::set_lean_layers(bool enabled) { g_lean_layers = enabled; }
bool mli::runtime::lean_layers() { return g_lean_layers; }

namespace {

using mli::runtime::Scratch;
inline Scratch scratch_for(int n_batch, int n_sequence, int dim) {
    return mli::runtime::attention_scratch(n_batch, n_sequence, dim);
}

inline void* stream() { return mli::runtime::compute_stream(); }
inline float* const* pages(const TensorFloatPoint& t) { return t.data(); }

}  // namespace

// ---- contiguous layout ------------------------------------------------------------------------------
void launch_fill_new_kt_v_cache(const TensorFloat& inp_embedding, const TensorInt& new_batch_idx,
                                const TensorInt& lengths, const TensorFloat& wk, const TensorFloat& wv,
                                TensorFloat& kt_cache, TensorFloat& v_cache, int n_new_items) {
    const auto& s = inp_embedding.shape();
    HIP_CHECK(mli_fill_new_kt_v_cache(inp_embedding.data(), new_batch_idx.data(), lengths.data(), wk.data(), wv.data(),
                                      kt_cache.data(), v_cache.data(), (int)s[0], (int)s[1], (int)s[2],
                                      (int)wk.shape()[1], n_new_items, stream()));
}

void launch_get_latest_kt_q_v(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorFloat& wk,
                              const TensorFloat& wq, const TensorFloat& wv, TensorFloat& kt_cache,
                              TensorFloat& v_cache, TensorFloat& q_output) {
    const auto& s = inp_embedding.shape();
    HIP_CHECK(mli_get_latest_kt_q_v(inp_embedding.data(), lengths.data(), wk.data(), wq.data(), wv.data(),
                                    kt_cache.data(), v_cache.data(), q_output.data(), (int)s[0], (int)s[1], (int)s[2],
                                    (int)wk.shape()[1], stream()));
}

void launch_qkt(const TensorFloat& q_output, const TensorFloat& kt_cache, const TensorInt& lengths,
                TensorFloat& qkt_output) {
    HIP_CHECK(mli_qkt(q_output.data(), kt_cache.data(), lengths.data(), qkt_output.data(), (int)q_output.shape()[0],
                      (int)kt_cache.shape()[2], (int)q_output.shape()[1], stream()));
}

void launch_softmax_in_place_with_lengths(TensorFloat& qkt_output, const TensorInt& lengths) {
    HIP_CHECK(mli_softmax_in_place_with_lengths(qkt_output.data(), lengths.data(), (int)qkt_output.shape()[0],
                                                (int)qkt_output.shape()[1], stream()));
}

void launch_softmax_v(const TensorFloat& softmax_result, const TensorFloat& v_cache, TensorFloat& attention_result,
                      const TensorInt& lengths) {
    const auto& s = v_cache.shape();
    const Scratch ws = scratch_for((int)s[0], (int)s[1], (int)s[2]);
    HIP_CHECK(mli_softmax_v(softmax_result.data(), v_cache.data(), lengths.data(), attention_result.data(), (int)s[0],
                            (int)s[1], (int)s[2], ws.ptr, ws.bytes, stream()));
}

void inference_self_attention(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorFloat& wk,
                              const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                              TensorFloat& kt_cache, TensorFloat& v_cache, TensorFloat& q_output,
                              TensorFloat& qkt_output, TensorFloat& attention_result, int n_new_items) {
    const auto& s = inp_embedding.shape();
    const int out_dim = (int)wk.shape()[1];
    const Scratch ws = scratch_for((int)s[0], (int)s[1], out_dim);
    HIP_CHECK(mli_inference_self_attention(inp_embedding.data(), lengths.data(), wk.data(), wq.data(), wv.data(),
                                           new_batch_idx.data(), kt_cache.data(), v_cache.data(), q_output.data(),
                                           qkt_output.data(), attention_result.data(), (int)s[0], (int)s[1], (int)s[2],
                                           out_dim, n_new_items, ws.ptr, ws.bytes, stream()));
}

void inference_self_attention_lean(const TensorFloat& inp_embedding, const TensorInt& lengths, const TensorFloat& wk,
                                   const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                                   TensorFloat& kt_cache, TensorFloat& v_cache, TensorFloat& q_output,
                                   TensorFloat& qkt_output, TensorFloat& attention_result, int n_new_items) {
    const auto& s = inp_embedding.shape();
    const int out_dim = (int)wk.shape()[1];
    const Scratch ws = scratch_for((int)s[0], (int)s[1], out_dim);
    const int rc = mli_self_attention_lean(inp_embedding.data(), lengths.data(), wk.data(), wq.data(), wv.data(),
                                           new_batch_idx.data(), kt_cache.data(), v_cache.data(), q_output.data(),
                                           attention_result.data(), (int)s[0], (int)s[1], (int)s[2], out_dim, n_new_items,
                                           ws.ptr, ws.bytes, stream());
    if (rc == MLI_ERR_BAD_ARG) {
        inference_self_attention(inp_embedding, lengths, wk, wq, wv, new_batch_idx, kt_cache, v_cache, q_output, qkt_output,
                                 attention_result, n_new_items);
        return;
    }
    HIP_CHECK(rc);
}

// ---- paged layout -----------------------------------------------------------------------------------
void launch_fill_new_k_v_cache_paged_attention(TensorFloatPoint page_table, const TensorInt& new_batch_idx,
                                               const TensorInt& lengths, const TensorFloat& wk,
                                               const TensorFloat& wv, int n_new_items, int n_sequence) {
    if (n_new_items == 0) return;
    assert(page_table.shape()[1] == (size_t)(n_sequence / PAGE_BLOCK_SIZE) && n_sequence % PAGE_BLOCK_SIZE == 0);
    assert(wk.shape()[0] == wk.shape()[1]);
    HIP_CHECK(mli_fill_new_k_v_cache_paged(pages(page_table), new_batch_idx.data(), lengths.data(), wk.data(),
                                           wv.data(), (int)page_table.shape()[0], n_sequence, (int)wk.shape()[0],
                                           n_new_items, stream()));
}

void launch_fill_new_k_v_cache_paged_attention_warp_tiling(TensorFloatPoint page_table,
                                                           const TensorInt& new_batch_idx, const TensorInt& lengths,
                                                           const TensorFloat& wk, const TensorFloat& wv,
                                                           int n_new_items, int n_sequence) {
    launch_fill_new_k_v_cache_paged_attention(page_table, new_batch_idx, lengths, wk, wv, n_new_items, n_sequence);
}

void launch_get_latest_k_q_v_paged_attention(TensorFloatPoint& page_table, const TensorInt& lengths,
                                             const TensorFloat& wk, const TensorFloat& wq, const TensorFloat& wv,
                                             TensorFloat& q_output, int n_sequence) {
    HIP_CHECK(mli_get_latest_k_q_v_paged(pages(page_table), lengths.data(), wk.data(), wq.data(), wv.data(),
                                         q_output.data(), (int)page_table.shape()[0], n_sequence, (int)wq.shape()[0],
                                         stream()));
}

void launch_get_latest_k_q_v_paged_attention_cublas(TensorFloatPoint& page_table, const TensorInt& lengths,
                                                    TensorFloat& /*latest_emb*/, const TensorFloat& wk,
                                                    const TensorFloat& wq, const TensorFloat& wv,
                                                    TensorFloat& q_output, TensorFloat& /*temp_placeholder*/,
                                                    GemmHandle& /*handle*/, int n_sequence) {
    launch_get_latest_k_q_v_paged_attention(page_table, lengths, wk, wq, wv, q_output, n_sequence);
}

void launch_qkt_paged_attention(const TensorFloat& q_output, const TensorFloatPoint& page_table,
                                const TensorInt& lengths, TensorFloat& qkt_output) {
    HIP_CHECK(mli_qkt_paged(q_output.data(), pages(page_table), lengths.data(), qkt_output.data(),
                            (int)q_output.shape()[0], (int)qkt_output.shape()[1], (int)q_output.shape()[1], stream()));
}

void launch_softmax_v_paged_attention(const TensorFloat& softmax_result, const TensorFloatPoint& page_table,
                                      TensorFloat& attention_result, const TensorInt& lengths) {
    const int B = (int)softmax_result.shape()[0], S = (int)softmax_result.shape()[1];
    const int D = (int)attention_result.shape()[1];
    const Scratch ws = scratch_for(B, S, D);
    HIP_CHECK(mli_softmax_v_paged(softmax_result.data(), pages(page_table), lengths.data(), attention_result.data(), B,
                                  S, D, ws.ptr, ws.bytes, stream()));
}

void paged_attention(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorFloat& wk,
                     const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                     TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result, int n_new_items,
                     int n_sequence) {
    const int B = (int)page_table.shape()[0], D = (int)wk.shape()[0];
    const Scratch ws = scratch_for(B, n_sequence, D);
    HIP_CHECK(mli_paged_attention(pages(page_table), lengths.data(), wk.data(), wq.data(), wv.data(),
                                  new_batch_idx.data(), q_output.data(), qkt_output.data(), attention_result.data(),
                                  B, n_sequence, D, n_new_items, ws.ptr, ws.bytes, stream()));
}

void launch_paged_prefill(const TensorFloat& emb_table, const TensorFloat& wpe, const TensorInt& inp,
                          TensorFloatPoint& page_table, const TensorInt& lengths, const TensorInt& new_item_indices,
                          const TensorFloat& wk, const TensorFloat& wv, int n_new_items) {
    if (n_new_items == 0) return;
    HIP_CHECK(mli_paged_prefill(emb_table.data(), wpe.data(), inp.data(), reinterpret_cast<void* const*>(pages(page_table)),
                                lengths.data(), new_item_indices.data(), wk.data(), wv.data(), (int)inp.shape()[0],
                                (int)inp.shape()[1], (int)emb_table.shape()[1], n_new_items, /*elem_bf16=*/0, stream()));
}

void launch_prefill(const TensorFloat& emb_table, const TensorFloat& wpe, const TensorInt& inp,
                    TensorFloat& inp_embedding, const TensorInt& lengths, const TensorInt& new_item_indices,
                    const TensorFloat& wk, const TensorFloat& wv, TensorFloat& kt_cache, TensorFloat& v_cache,
                    int n_new_items) {
    if (n_new_items == 0) return;
    const auto& s = inp_embedding.shape();
    HIP_CHECK(mli_prefill(emb_table.data(), wpe.data(), inp.data(), inp_embedding.data(), lengths.data(),
                          new_item_indices.data(), wk.data(), wv.data(), kt_cache.data(), v_cache.data(), (int)s[0],
                          (int)s[1], (int)s[2], (int)wk.shape()[1], n_new_items, stream()));
}

// EXTENSION: the composition without the scores (include/mli_kernels.h: mli_paged_attention_lean).  Rows too wide for
// the single-pass kernel take the materialising composition into the caller's qkt_output scratch instead.
void paged_attention_lean(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorFloat& wk,
                          const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                          TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result,
                          int n_new_items, int n_sequence) {
    const int B = (int)page_table.shape()[0], D = (int)wk.shape()[0];
    const Scratch ws = scratch_for(B, n_sequence, D);
    const int rc = mli_paged_attention_lean(reinterpret_cast<void* const*>(pages(page_table)), lengths.data(), wk.data(),
                                            wq.data(), wv.data(), new_batch_idx.data(), q_output.data(),
                                            attention_result.data(), B, n_sequence, D, n_new_items, /*elem_bf16=*/0,
                                            ws.ptr, ws.bytes, stream());
    if (rc == MLI_ERR_BAD_ARG && D > 2048 && D % 4 == 0) {
        paged_attention(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, qkt_output, attention_result,
                        n_new_items, n_sequence);
        return;
    }
    HIP_CHECK(rc);
}

void paged_attention_with_cublas(TensorFloatPoint& page_table, const TensorInt& lengths, const TensorFloat& wk,
                                 const TensorFloat& wq, const TensorFloat& wv, const TensorInt& new_batch_idx,
                                 TensorFloat& q_output, TensorFloat& qkt_output, TensorFloat& attention_result,
                                 TensorFloat& /*latest_emb*/, TensorFloat& /*temp_placeholder*/, int n_new_items,
                                 int n_sequence, GemmHandle& /*handle*/) {
    paged_attention(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, qkt_output, attention_result,
                    n_new_items, n_sequence);
}

// ---- encoder / decoder ------------------------------------------------------------------------------
void launch_inference_optimized_encoder_kernel(const float* emb_table, const float* wpe, const int* inp,
                                               float* inp_embedding, const int* lengths,
                                               const int* new_item_indices, int batch_size, int n_sequence,
                                               int embedding_dim, int n_new_items) {
    HIP_CHECK(mli_inference_optimized_encoder(emb_table, wpe, inp, inp_embedding, lengths, new_item_indices,
                                              batch_size, n_sequence, embedding_dim, n_new_items, stream()));
}

void launch_paged_attention_encoder_kernel(const float* emb_table, const float* wpe, const int* inp,
                                           float** page_table, const int* lengths, const int* new_item_indices,
                                           int batch_size, int n_sequence, int embedding_dim, int n_new_items) {
    HIP_CHECK(mli_paged_attention_encoder(emb_table, wpe, inp, page_table, lengths, new_item_indices, batch_size,
                                          n_sequence, embedding_dim, n_new_items, stream()));
}

void launch_decoder(const TensorFloat& batch_result, const TensorFloat& emb_table, TensorFloat& emb_score,
                    const TensorFloat& wpe_table, TensorFloat& inp_embedding, TensorInt& lengths,
                    TensorInt& decoder_result) {
    HIP_CHECK(mli_decoder(batch_result.data(), emb_table.data(), emb_score.data(), wpe_table.data(),
                          inp_embedding.data(), lengths.data(), decoder_result.data(), (int)batch_result.shape()[0],
                          (int)emb_table.shape()[0], (int)wpe_table.shape()[0], (int)batch_result.shape()[1],
                          stream()));
}

void launch_paged_attention_decoder_multi_rounds(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                                 TensorFloat& emb_score, const TensorFloat& wpe_table,
                                                 TensorFloatPoint& page_table, TensorInt& lengths,
                                                 TensorInt& decoder_result, int i_decoder) {
    const int n_results = decoder_result.shape().size() == 2 ? (int)decoder_result.shape()[1] : 1;
    HIP_CHECK(mli_paged_decoder_multi_rounds(batch_result.data(), emb_table.data(), emb_score.data(),
                                             wpe_table.data(), pages(page_table), lengths.data(),
                                             decoder_result.data(), (int)batch_result.shape()[0],
                                             (int)emb_table.shape()[0], (int)wpe_table.shape()[0],
                                             (int)batch_result.shape()[1], n_results, i_decoder, stream()));
}

// EXTENSION: decoder head with the argmax as the logits GEMM's epilogue (mli_decoder_fused /
// mli_paged_decoder_fused).  emb_score is only lent as scratch: [n_batch, n_vocab] floats hold the
// [n_batch, ceil(n_vocab / 64)] (value, index) pairs with room to spare; its contents are unspecified afterwards.
void launch_decoder_fused(const TensorFloat& batch_result, const TensorFloat& emb_table, TensorFloat& emb_score,
                          const TensorFloat& wpe_table, TensorFloat& inp_embedding, TensorInt& lengths,
                          TensorInt& decoder_result) {
    HIP_CHECK(mli_decoder_fused(batch_result.data(), emb_table.data(), wpe_table.data(), inp_embedding.data(),
                                lengths.data(), decoder_result.data(), (int)batch_result.shape()[0],
                                (int)emb_table.shape()[0], (int)wpe_table.shape()[0], (int)batch_result.shape()[1],
                                emb_score.data(), emb_score.get_total_size() * sizeof(float), stream()));
}

void launch_paged_attention_decoder_fused(const TensorFloat& batch_result, const TensorFloat& emb_table,
                                          TensorFloat& emb_score, const TensorFloat& wpe_table,
                                          TensorFloatPoint& page_table, TensorInt& lengths, TensorInt& decoder_result,
                                          int i_decoder) {
    const int n_results = decoder_result.shape().size() == 2 ? (int)decoder_result.shape()[1] : 1;
    HIP_CHECK(mli_paged_decoder_fused(batch_result.data(), emb_table.data(), wpe_table.data(),
                                      reinterpret_cast<void* const*>(pages(page_table)), lengths.data(),
                                      decoder_result.data(), (int)batch_result.shape()[0], (int)emb_table.shape()[0],
                                      (int)wpe_table.shape()[0], (int)batch_result.shape()[1], n_results, i_decoder,
                                      /*elem_bf16=*/0, emb_score.data(), emb_score.get_total_size() * sizeof(float),
                                      stream()));
}

void launch_paged_attention_cublas_decoder_multi_rounds(const TensorFloat& batch_result,
                                                        const TensorFloat& emb_table, TensorFloat& emb_score,
                                                        const TensorFloat& wpe_table, TensorFloatPoint& page_table,
                                                        TensorInt& lengths, TensorInt& decoder_result, int i_decoder,
                                                        GemmHandle& /*handle*/) {
    launch_paged_attention_decoder_multi_rounds(batch_result, emb_table, emb_score, wpe_table, page_table, lengths,
                                                decoder_result, i_decoder);
}

void launch_clone_inp_embedding_k_v_cache(float** page_table, const float* inp_embedding, const float* kt_cache,
                                          const float* v_cache, const int* lengths, int n_batch, int n_sequence,
                                          int emb_dim) {
    HIP_CHECK(mli_clone_inp_embedding_k_v_cache(page_table, inp_embedding, kt_cache, v_cache, lengths, n_batch,
                                                n_sequence, emb_dim, stream()));
}
