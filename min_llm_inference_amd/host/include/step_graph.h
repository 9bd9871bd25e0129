// hipGraph replay of a pure decode forward (n_new_items == 0).  EXTENSION -- SURVEY 8(f) row 3 ("hipGraph capture of
// the step"): the reference's forward issues its launches one by one (src/inference_model.cpp:56-81); a decode forward
// touches device state (lengths, page table, pages) only through pointers, so the same launches can be recorded once
// and replayed with ONE host call per forward.
//
// Off by default (mli::runtime::set_step_graphs): on this stack a replay costs the GPU 3-5 us more per forward than the
// same launches issued eagerly from C++ (measured), so it pays only where the HOST is the bottleneck.  Needs
// a real compute stream (the legacy default stream cannot be captured): an engine with a private stream.
#pragma once

#include <vector>

class StepGraph {
public:
    StepGraph() = default;
    ~StepGraph();
    StepGraph(const StepGraph&) = delete;
    StepGraph& operator=(const StepGraph&) = delete;
    StepGraph(StepGraph&& other) noexcept;

    // Runs `body` -- the launches of one decode forward over the buffers named by `key` -- or its recorded replay.
    // The first call with a key runs eagerly (scratch may still be allocated then), the second records, later ones
    // replay; a different key (other tensors, other stream) starts over.
    template <class Body>
    void run(const std::vector<const void*>& key, Body&& body) {
        if (!begin(key)) {          // eager (graphs off, no capturable stream, or warm-up), or replayed already
            if (!replayed_) body();
            return;
        }
        try {
            body();                  // recorded, not executed
        } catch (...) {
            abandon();
            throw;
        }
        finish();                    // instantiate + first launch
    }

private:
    bool begin(const std::vector<const void*>& key);   // true = capture has started
    void finish();
    void abandon() noexcept;
    void reset() noexcept;
    void* exec_ = nullptr;
    std::vector<const void*> key_;
    int seen_ = 0;
    bool replayed_ = false;
};
