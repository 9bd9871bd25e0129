// Per-thread execution context: which GPU this thread drives and which stream kernels go to.
// The reference is single-GPU and launches everything on the legacy default stream; that stays the
// default here (compute_stream() == nullptr).  A multi-GPU host creates one engine per GPU, each on its
// own thread or process, and calls use_device() first.
#pragma once

#include <cstddef>

namespace mli {
namespace runtime {

void use_device(int ordinal);          // hipSetDevice for the calling thread
int current_device();
void set_compute_stream(void* stream); // hipStream_t as void*; nullptr = legacy default stream
void* compute_stream();
void* create_stream();                  // a non-blocking stream for one engine (several engines per GPU overlap)
void destroy_stream(void* stream) noexcept;
void synchronize();                    // hipStreamSynchronize(compute stream) / device sync for the default stream

// Device scratch of the split-sequence attention kernels (mli_attention_workspace_bytes): one buffer per device,
// grown on demand, owned by the host library because the C ABI never allocates.
struct Scratch {
    void* ptr;
    size_t bytes;
};
Scratch attention_scratch(int n_batch, int n_sequence, int emb_dim);
void release_attention_scratch(void* stream) noexcept;  // frees the buffers tied to a stream that is going away

// The three switches below are PER THREAD, like the device and the compute stream: they apply to what the calling thread
// runs afterwards (an engine is driven by one thread at a time; engines created through the C ABI carry their own values,
// mli_engine.h).
//
// What the layers run (default true): the LEAN compositions -- attention without materialising scores /
// probabilities in the layer's qkt_output scratch, decoder head with the argmax as the logits GEMM's epilogue instead of
// materialising emb_score.  Outputs a caller of forward() can observe (attention_result, tokens, lengths, pages) are
// bit-identical either way; false reproduces the reference's launch sequence, scratch contents included.
void set_lean_layers(bool enabled);
bool lean_layers();

// Which loop start_paged_attention_*_inference_engine runs (default false): the pipelined loop
// (pipelined_engine.h: the host one step behind the GPU, same tokens per item) wherever it applies -- up to
// PAGE_BLOCK_SIZE / 2 forward rounds, no length-reset quirk --, or, with true, always the reference's sequential order
// forward -> process_decoder_result -> allocate_or_free_memory_blocks_if_needed -> insert_new_items.
void set_sequential_engine_loop(bool enabled);
bool sequential_engine_loop();

// hipGraph replay of decode forwards (step_graph.h); default false.
void set_step_graphs(bool enabled);
bool step_graphs();

// roctx ranges around engine phases (the reference wraps them in NVTX ranges, src/inferencer.cpp:55-82)
void range_push(const char* name);
void range_pop();

}  // namespace runtime
}  // namespace mli
