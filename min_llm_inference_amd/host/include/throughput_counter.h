// The metric: tokens appended by process_decoder_result divided by wall time, host scheduling and copies
// included (reference include/throughput_counter.h, src/throughput_counter.cpp:8-35).
#pragma once

#include <chrono>

class ThroughputCounter {
public:
    ThroughputCounter();
    void print_throughput();                       // "Total tokens: N, seconds: S, throughput: N/S"
    void start_record();
    void add_record_if_recording(int new_tokens);
    // extensions: read / reset the totals (the C ABI reports them instead of parsing stdout)
    long long total_tokens() const { return total_tokens_; }
    double seconds() const { return micro_seconds_ * 1e-6; }
    void reset();

private:
    long long total_tokens_;
    long long micro_seconds_;
    std::chrono::time_point<std::chrono::steady_clock> last_timestamp_;
    bool in_recording_;
};

ThroughputCounter& get_global_throughput_counter();

// Extension: route the calling thread's records to `counter` (nullptr = back to the process-wide one), so several
// engines can run in one process, one thread each, without sharing the reference's singleton.
void set_thread_throughput_counter(ThroughputCounter* counter);
