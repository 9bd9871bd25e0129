// Error convention, small helpers and the page-layout rule (reference include/utils.h, src/utils.cpp).
#pragma once

#include <cstddef>
#include <cstdint>

// Non-zero C-ABI / HIP status -> print "[HIP ERROR] at file ..." and throw std::runtime_error("Hip Failure")
// (the reference's cuda_check prints and throws std::runtime_error("Cuda Failure"), src/utils.cpp:5-11).
void hip_check(int status, const char* file, int line);
#define HIP_CHECK(call) hip_check(static_cast<int>(call), __FILE__, __LINE__)

// Checks the sticky launch error; with -DUSE_SYNC_HIP_CHECK it synchronises the device first, like the
// reference's DEBUG_MODE build (include/utils.h:12-26).
void hip_check_last(const char* file, int line);
#define HIP_CHECK_LAST() hip_check_last(__FILE__, __LINE__)

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Float offset of element (i_sequence, segment, i_dim) inside its page block (include/utils.h:37,43,52,59).
inline std::int64_t page_block_offset(int i_sequence, int emb_dim, int page_block_size, int i_dim, int emb_offset) {
    return static_cast<std::int64_t>(i_sequence % page_block_size) * emb_dim * 3 +
           static_cast<std::int64_t>(emb_offset) * emb_dim + i_dim;
}

// Index of the page-table entry that holds (i_batch, i_sequence).
inline std::int64_t page_table_index(int i_batch, int n_sequence, int i_sequence, int page_block_size) {
    return static_cast<std::int64_t>(i_batch) * (n_sequence / page_block_size) + i_sequence / page_block_size;
}

class NonCopyableNonClonable {
protected:
    NonCopyableNonClonable() = default;
    ~NonCopyableNonClonable() = default;
    NonCopyableNonClonable(NonCopyableNonClonable&&) noexcept = default;
    NonCopyableNonClonable& operator=(NonCopyableNonClonable&&) noexcept = default;

public:
    NonCopyableNonClonable(const NonCopyableNonClonable&) = delete;
    NonCopyableNonClonable& operator=(const NonCopyableNonClonable&) = delete;
};

// contiguous -> paged copy used by the parity tests (reference include/utils.h:101-103)
void launch_clone_inp_embedding_k_v_cache(float** page_table, const float* inp_embedding, const float* kt_cache,
                                          const float* v_cache, const int* lengths, int n_batch, int n_sequence,
                                          int emb_dim);
