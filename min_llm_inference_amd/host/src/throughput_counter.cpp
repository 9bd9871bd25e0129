#include "throughput_counter.h"

#include <iostream>

ThroughputCounter::ThroughputCounter() : total_tokens_(0), micro_seconds_(0), in_recording_(false) {}

void ThroughputCounter::reset() {
    total_tokens_ = 0;
    micro_seconds_ = 0;
    in_recording_ = false;
}

void ThroughputCounter::start_record() {
    if (in_recording_) return;
    last_timestamp_ = std::chrono::steady_clock::now();
    in_recording_ = true;
}

void ThroughputCounter::add_record_if_recording(int new_tokens) {
    if (!in_recording_) return;
    const auto now = std::chrono::steady_clock::now();
    micro_seconds_ += std::chrono::duration_cast<std::chrono::microseconds>(now - last_timestamp_).count();
    total_tokens_ += new_tokens;
    last_timestamp_ = now;  // keep recording: the next interval starts here
}

void ThroughputCounter::print_throughput() {
    const double s = seconds();
    std::cout << "Total tokens: " << total_tokens_ << ", seconds: " << s
              << ", throughput: " << (s > 0 ? total_tokens_ / s : 0.0) << std::endl;
}

namespace {
thread_local ThroughputCounter* t_counter = nullptr;
}

void set_thread_throughput_counter(ThroughputCounter* counter) { t_counter = counter; }

ThroughputCounter& get_global_throughput_counter() {
    static ThroughputCounter counter;
    return t_counter ? *t_counter : counter;
}
