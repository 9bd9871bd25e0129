// HIP implementation of the Tensor memory backend (memory.h) and of the per-thread runtime context.
//   sync  flavour: hipMalloc / hipHostMalloc / hipMemcpy                      (reference tensor.hpp:272-322)
//   async flavour: hipMallocAsync / hipMemcpyAsync on one transfer stream per device plus a readiness
//                  event per block that pointer() waits on                      (reference tensor.hpp:182-269)
#include <hip/hip_runtime.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <atomic>
#include <cstdio>
#include <map>
#include <mutex>
#include <stdexcept>

#include "memory.h"
#include "runtime.h"
#include "utils.h"

void hip_check(int status, const char* file, int line) {
    if (status != 0) {
        const char* msg = status > 0 ? hipGetErrorString(static_cast<hipError_t>(status)) : "invalid argument to the mli C ABI";
        std::printf("[HIP ERROR] at file %s:%d:\n%s (%d)\n", file, line, msg, status);
        throw std::runtime_error("Hip Failure");
    }
}

void hip_check_last(const char* file, int line) {
#ifdef USE_SYNC_HIP_CHECK
    hip_check(static_cast<int>(hipDeviceSynchronize()), file, line);
#endif
    hip_check(static_cast<int>(hipGetLastError()), file, line);
}

namespace mli {
namespace runtime {

namespace {
thread_local void* t_stream = nullptr;
}

void use_device(int ordinal) { HIP_CHECK(hipSetDevice(ordinal)); }

int current_device() {
    int d = 0;
    HIP_CHECK(hipGetDevice(&d));
    return d;
}

void set_compute_stream(void* stream) { t_stream = stream; }
void* compute_stream() { return t_stream; }

void synchronize() {
    if (t_stream) HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(t_stream)));
    else HIP_CHECK(hipDeviceSynchronize());
}

void* create_stream() {
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    return s;
}
void destroy_stream(void* stream) noexcept {
    if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
}

void range_push(const char* name) { roctxRangePushA(name); }
void range_pop() { roctxRangePop(); }

}  // namespace runtime

namespace mem {

namespace {
std::atomic<int> g_default_mode{static_cast<int>(Mode::Sync)};
}
void set_process_default_mode(Mode mode) { g_default_mode.store(static_cast<int>(mode)); }
Mode process_default_mode() { return static_cast<Mode>(g_default_mode.load()); }

struct Block {
    void* ptr = nullptr;
    std::size_t bytes = 0;
    Space space = Space::Host;
    Mode mode = Mode::Sync;
    int device = 0;
    hipEvent_t ready = nullptr;  // async flavour: recorded after the last operation that touched ptr
    bool pending = false;
};

namespace {

// One transfer stream per device for the async flavour (created on first use, never destroyed --
// the reference leaks its stream the same way, tensor.hpp:123-130).
hipStream_t transfer_stream(int device) {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    HIP_CHECK(hipStreamCreate(&s));
    streams[device] = s;
    return s;
}

void mark_pending(Block* b) {
    if (b->ready == nullptr) HIP_CHECK(hipEventCreateWithFlags(&b->ready, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(b->ready, transfer_stream(b->device)));
    b->pending = true;
}

void wait_ready(Block* b) {
    if (b->pending) {
        HIP_CHECK(hipEventSynchronize(b->ready));
        b->pending = false;
    }
}

hipMemcpyKind kind_of(Space dst, Space src) {
    if (dst == Space::Host) return src == Space::Host ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    return src == Space::Host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
}

}  // namespace

Block* acquire(std::size_t bytes, Space space, Mode mode) {
    Block* b = new Block;
    b->bytes = bytes;
    b->space = space;
    b->mode = mode;
    b->device = runtime::current_device();
    const std::size_t n = bytes ? bytes : 1;
    try {
        if (space == Space::Host) {
            HIP_CHECK(hipHostMalloc(&b->ptr, n, hipHostMallocDefault));  // pinned, as the reference's cudaHostAlloc
        } else if (mode == Mode::Sync) {
            HIP_CHECK(hipMalloc(&b->ptr, n));
        } else {
            HIP_CHECK(hipMallocAsync(&b->ptr, n, transfer_stream(b->device)));
            mark_pending(b);
        }
    } catch (...) {
        delete b;
        throw;
    }
    return b;
}

void release(Block* b) noexcept {
    if (b == nullptr) return;
    if (b->space == Space::Host) {
        if (b->pending) (void)hipEventSynchronize(b->ready);
        (void)hipHostFree(b->ptr);
    } else if (b->mode == Mode::Sync) {
        (void)hipFree(b->ptr);
    } else {
        (void)hipFreeAsync(b->ptr, transfer_stream(b->device));
    }
    if (b->ready) (void)hipEventDestroy(b->ready);
    delete b;
}

void* pointer(Block* b) {
    wait_ready(b);
    return b->ptr;
}

void copy(Block* dst, const Block* src, std::size_t byte_offset, std::size_t bytes) {
    if (bytes == 0) return;
    Block* s = const_cast<Block*>(src);
    const hipMemcpyKind kind = kind_of(dst->space, src->space);
    char* d = static_cast<char*>(dst->ptr) + byte_offset;
    const char* f = static_cast<const char*>(s->ptr) + byte_offset;
    if (dst->mode == Mode::Sync) {
        // With a private compute stream the copy must be ordered with THAT stream's kernels (hipMemcpy only orders
        // with the legacy default stream and would serialise against every other engine in the process).
        hipStream_t cs = static_cast<hipStream_t>(runtime::compute_stream());
        if (cs == nullptr) {
            HIP_CHECK(hipMemcpy(d, f, bytes, kind));
        } else {
            HIP_CHECK(hipMemcpyAsync(d, f, bytes, kind, cs));
            HIP_CHECK(hipStreamSynchronize(cs));
        }
    } else {
        wait_ready(s);
        wait_ready(dst);
        hipStream_t ts = transfer_stream(dst->device);
        // The transfer stream is a blocking stream: it orders itself with the LEGACY default stream, which is where
        // kernels run unless the engine has a private (non-blocking) compute stream.  With one, nothing is implicit:
        // the copy must not start before the kernels queued so far have run (a D2H of their results, or an H2D into
        // a buffer they still read), so the transfer stream first waits for that point of the compute stream.
        if (hipStream_t cs = static_cast<hipStream_t>(runtime::compute_stream())) {
            // one event per (thread, device): an event belongs to the device that was current when it was created, and a
            // thread may drive engines on several devices (row-sharded replicas in one process)
            constexpr int kMaxOrderedDevices = 64;
            thread_local hipEvent_t ordered_of[kMaxOrderedDevices] = {};
            if (dst->device < 0 || dst->device >= kMaxOrderedDevices) throw std::runtime_error("device ordinal out of range");
            hipEvent_t& ordered = ordered_of[dst->device];
            if (ordered == nullptr) HIP_CHECK(hipEventCreateWithFlags(&ordered, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(ordered, cs));
            HIP_CHECK(hipStreamWaitEvent(ts, ordered, 0));
        }
        HIP_CHECK(hipMemcpyAsync(d, f, bytes, kind, ts));
        mark_pending(dst);
    }
}

namespace {
// the updates travel in the kernel arguments: no staging buffer, no copy, one launch per 128 entries
constexpr int kScatterPerLaunch = 128;
struct ScatterArgs {
    long long index[kScatterPerLaunch];
    unsigned long long value[kScatterPerLaunch];
};
template <typename T>
__global__ void scatter_kernel(T* dst, ScatterArgs a, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[a.index[i]] = static_cast<T>(a.value[i]);
}

template <typename T>
void scatter_impl(Block* dst, const long long* index, const T* value, std::size_t n) {
    if (dst->space == Space::Host) {
        for (std::size_t i = 0; i < n; ++i) static_cast<T*>(dst->ptr)[index[i]] = value[i];
        return;
    }
    wait_ready(dst);
    hipStream_t st = static_cast<hipStream_t>(runtime::compute_stream());
    for (std::size_t done = 0; done < n; done += kScatterPerLaunch) {
        ScatterArgs a;
        const int m = static_cast<int>(n - done < kScatterPerLaunch ? n - done : kScatterPerLaunch);
        for (int i = 0; i < m; ++i) {
            a.index[i] = index[done + i];
            a.value[i] = value[done + i];
        }
        hipLaunchKernelGGL(scatter_kernel<T>, dim3(1), dim3(kScatterPerLaunch), 0, st, static_cast<T*>(dst->ptr), a, m);
        HIP_CHECK(hipGetLastError());
    }
}
}  // namespace

struct Marker {
    hipEvent_t event = nullptr;
};

Marker* create_marker() {
    Marker* m = new Marker;
    HIP_CHECK(hipEventCreateWithFlags(&m->event, hipEventDisableTiming));
    return m;
}
void destroy_marker(Marker* m) noexcept {
    if (m == nullptr) return;
    (void)hipEventDestroy(m->event);
    delete m;
}
void record_marker(Marker* m) { HIP_CHECK(hipEventRecord(m->event, static_cast<hipStream_t>(runtime::compute_stream()))); }
void wait_marker(Marker* m) { HIP_CHECK(hipEventSynchronize(m->event)); }

void copy_async(Block* dst, const Block* src, std::size_t byte_offset, std::size_t bytes) {
    if (bytes == 0) return;
    if (dst->mode != Mode::Sync || src->mode != Mode::Sync)
        throw std::runtime_error("copy_async: sync-flavour blocks only");
    HIP_CHECK(hipMemcpyAsync(static_cast<char*>(dst->ptr) + byte_offset,
                             static_cast<const char*>(src->ptr) + byte_offset, bytes, kind_of(dst->space, src->space),
                             static_cast<hipStream_t>(runtime::compute_stream())));
}

void scatter8(Block* dst, const long long* index, const unsigned long long* value, std::size_t n) {
    scatter_impl<unsigned long long>(dst, index, value, n);
}
void scatter4(Block* dst, const long long* index, const unsigned int* value, std::size_t n) {
    scatter_impl<unsigned int>(dst, index, value, n);
}

Space space_of(const Block* b) { return b->space; }
Mode mode_of(const Block* b) { return b->mode; }
std::size_t size_of(const Block* b) { return b->bytes; }

}  // namespace mem
}  // namespace mli
