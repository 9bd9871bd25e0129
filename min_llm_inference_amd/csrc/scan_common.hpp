// Shared by the single-pass scan kernels (attention_fused.hip: chunked grid; attention_stream.hip: equal page shares):
// element types of the page rows, the 16-byte lane load and a compile-time loop.
#pragma once

#include <type_traits>

#include "device_common.hpp"

namespace mli {

typedef uint32_t fu_u32x4 __attribute__((ext_vector_type(4)));
typedef const fu_u32x4 __attribute__((address_space(1)))* fu_gu4_ptr;

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

template <bool NT>
__device__ __forceinline__ fu_u32x4 fu_ldg(const void* p) {
    if (NT) return __builtin_nontemporal_load((fu_gu4_ptr)(p));
    return *(fu_gu4_ptr)(p);
}

// one 16-byte lane load holds EPL elements
struct ElemF32 {
    static constexpr int EPL = 4;
    static constexpr int kBytes = 4;
    static __device__ __forceinline__ void unpack(const fu_u32x4& r, float (&f)[4]) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
};
struct ElemBF16 {
    static constexpr int EPL = 8;
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ void unpack(const fu_u32x4& r, float (&f)[8]) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
};

}  // namespace mli
