// TEST DOUBLE for the Tensor memory backend (min_llm_inference_amd/host/include/memory.h): plain malloc /
// memcpy so that the host scheduler can be unit-tested in a container without a GPU.  Linked only into
// tests/cpp binaries -- never into libmli_hip.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "memory.h"
#include "runtime.h"
#include "utils.h"

void hip_check(int status, const char* file, int line) {
    if (status != 0) {
        std::printf("[HIP ERROR] at file %s:%d: status %d\n", file, line, status);
        throw std::runtime_error("Hip Failure");
    }
}
void hip_check_last(const char*, int) {}

namespace mli {
namespace mem {

static Mode g_default_mode = Mode::Sync;
void set_process_default_mode(Mode mode) { g_default_mode = mode; }
Mode process_default_mode() { return g_default_mode; }

struct Block {
    void* ptr;
    std::size_t bytes;
    Space space;
    Mode mode;
};

Block* acquire(std::size_t bytes, Space space, Mode mode) {
    Block* b = new Block{std::malloc(bytes ? bytes : 1), bytes, space, mode};
    if (!b->ptr) throw std::bad_alloc();
    return b;
}
void release(Block* b) noexcept {
    if (!b) return;
    std::free(b->ptr);
    delete b;
}
void* pointer(Block* b) { return b->ptr; }
void copy(Block* dst, const Block* src, std::size_t off, std::size_t bytes) {
    if (off + bytes > dst->bytes || off + bytes > src->bytes) throw std::runtime_error("copy out of range");
    std::memcpy(static_cast<char*>(dst->ptr) + off, static_cast<const char*>(src->ptr) + off, bytes);
}
void scatter8(Block* dst, const long long* index, const unsigned long long* value, std::size_t n) {
    for (std::size_t i = 0; i < n; ++i) {
        if (static_cast<std::size_t>(index[i]) * 8 + 8 > dst->bytes) throw std::runtime_error("scatter out of range");
        static_cast<unsigned long long*>(dst->ptr)[index[i]] = value[i];
    }
}
void scatter4(Block* dst, const long long* index, const unsigned int* value, std::size_t n) {
    for (std::size_t i = 0; i < n; ++i) {
        if (static_cast<std::size_t>(index[i]) * 4 + 4 > dst->bytes) throw std::runtime_error("scatter out of range");
        static_cast<unsigned int*>(dst->ptr)[index[i]] = value[i];
    }
}
void copy_async(Block* dst, const Block* src, std::size_t off, std::size_t bytes) { copy(dst, src, off, bytes); }
struct Marker {};
Marker* create_marker() { return new Marker; }
void destroy_marker(Marker* m) noexcept { delete m; }
void record_marker(Marker*) {}
void wait_marker(Marker*) {}
Space space_of(const Block* b) { return b->space; }
Mode mode_of(const Block* b) { return b->mode; }
std::size_t size_of(const Block* b) { return b->bytes; }

}  // namespace mem
namespace runtime {  // what the engine loops use of runtime.h: nothing to do without a device
void range_push(const char*) {}
void range_pop() {}
void synchronize() {}
}  // namespace runtime
}  // namespace mli
