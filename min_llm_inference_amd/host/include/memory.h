// Memory backend behind Tensor<T>.  Product builds link host/src/memory_hip.cpp (hipMalloc /
// hipHostMalloc / hipMemcpy[Async] on the engine's device); the CPU tests of the host scheduler link a
// plain-malloc test double instead (tests/cpp/memory_host_double.cpp) -- a link-time substitution, the
// product library contains no CPU fallback.
#pragma once

#include <cstddef>

namespace mli {
namespace mem {

enum class Space { Host, Device };
enum class Mode { Sync, Async };

struct Block;  // opaque; owns one allocation (+ the readiness event of the async flavour)

// The flavour a Tensor gets when its constructor is not told one.  The reference fixes it per BUILD
// (-DUSE_ASYNC_ALLOC -> DEFAULT_ALLOC_METHOD, include/tensor.hpp:21-25) and runs its whole test suite under both
// (Makefile:20-30); here the library is built once, so the flavour a host compiled with -DDEFAULT_ALLOC_METHOD=1 asks for is
// registered at load time (tensor.hpp) and every tensor of the process -- the library's own scratch included -- follows it.
void set_process_default_mode(Mode mode);
Mode process_default_mode();

Block* acquire(std::size_t bytes, Space space, Mode mode);
void release(Block* block) noexcept;
void* pointer(Block* block);                       // async blocks wait for their last copy first
void copy(Block* dst, const Block* src, std::size_t byte_offset, std::size_t bytes);  // same offset both sides
// dst holds 8-byte (scatter8) or 4-byte (scatter4) elements: element index[i] = value[i] for i < n (index / value
// are host arrays, consumed before the call returns).  Ordered like a kernel on the compute stream (after earlier
// work, before later work); returns without waiting for it.
void scatter8(Block* dst, const long long* index, const unsigned long long* value, std::size_t n);
void scatter4(Block* dst, const long long* index, const unsigned int* value, std::size_t n);

// Stream-ordered copy that does not block the host (sync-flavour blocks only; the host side must be pinned, which
// every Space::Host block is).  The source must stay unchanged, and a host destination unread, until a marker
// recorded after the copy has been waited for.
void copy_async(Block* dst, const Block* src, std::size_t byte_offset, std::size_t bytes);

// Marker = a point in the compute stream's work the host can wait for.
struct Marker;
Marker* create_marker();
void destroy_marker(Marker* marker) noexcept;
void record_marker(Marker* marker);   // after everything queued on the compute stream so far
void wait_marker(Marker* marker);     // blocks the host until that point has executed
Space space_of(const Block* block);
Mode mode_of(const Block* block);
std::size_t size_of(const Block* block);

}  // namespace mem
}  // namespace mli
