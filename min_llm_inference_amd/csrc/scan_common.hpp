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

// one 16-byte lane load holds EPL elements.  dot: s += q . (the load's elements); axpy: acc += p * (the load's elements)
template <class E>
struct ElemMath {
    static __device__ __forceinline__ void dot(const fu_u32x4& r, const float (&q)[E::EPL], float& s) {
        float f[E::EPL];
        E::unpack(r, f);
#pragma unroll
        for (int e = 0; e < E::EPL; ++e) s = fmaf(q[e], f[e], s);
    }
    static __device__ __forceinline__ void axpy(const fu_u32x4& r, float p, float (&acc)[E::EPL]) {
        float f[E::EPL];
        E::unpack(r, f);
#pragma unroll
        for (int e = 0; e < E::EPL; ++e) acc[e] = fmaf(p, f[e], acc[e]);
    }
};

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

// EXTENSION, opt-in (MLI_ELEM_FP8): OCP e4m3 page elements, 16 per 16-byte lane load; decoded by v_cvt_pk_f32_fp8
// (gfx950 decodes the OCP encodings, not MI300's fnuz ones: tools/fp8_probe.hip prints all 256)
struct ElemFP8 {
    static constexpr int EPL = 16;
    static constexpr int kBytes = 1;
    static __device__ __forceinline__ void unpack(const fu_u32x4& r, float (&f)[16]) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], false);
            const f32x2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(w[i], true);
            f[4 * i] = lo.x; f[4 * i + 1] = lo.y; f[4 * i + 2] = hi.x; f[4 * i + 3] = hi.y;
        }
    }
};

// ---- several token rows per load instruction (RPI) -------------------------------------------------------------------
// A 16-byte lane load moves 1 KiB per wave.  A row narrower than that (fp8 at emb_dim 512 is 512 bytes) would leave half
// of every load instruction, and half of the lanes' arithmetic, idle -- and 8-byte lane loads run at 0.54-0.70 x the
// 16-byte rate (MI355X_MICROARCH.md) -- so one instruction covers RPI = 2 or 4 consecutive token slots of the page: lane l
// holds unit l % (64 / RPI) of slot RPI * t + l / (64 / RPI) for instruction t.  The reduction of the partial scores is
// wave_reduce16's butterfly entered one or two steps later (the exchanges between lane groups are the ones skipped: the
// groups hold different slots), a slot's probability is fetched per lane group, and the groups' partial outputs are
// summed when the wave parks its state.
template <int RPI>
__device__ __forceinline__ int rpi_slot_of_lane(int lane) {   // the slot whose score the lane holds after the reduction
    if constexpr (RPI == 1) return (lane >> 2) & 15;
    else if constexpr (RPI == 2) return 2 * ((lane >> 2) & 7) + (lane >> 5);
    else return 4 * ((lane >> 2) & 3) + (lane >> 4);
}
template <int RPI>
__device__ __forceinline__ constexpr int rpi_lane_of_slot(int slot) {   // a lane that holds the slot's score
    return RPI == 1 ? 4 * slot : RPI == 2 ? 32 * (slot & 1) + 4 * (slot >> 1) : 16 * (slot & 3) + 4 * (slot >> 2);
}
// v[i]: partial score of slot RPI * i + (lane group) for this lane's unit; returns the slot's full score in every lane
// rpi_slot_of_lane names
template <int RPI>
__device__ __forceinline__ float rpi_reduce(float (&v)[16 / RPI], int lane) {
    if constexpr (RPI == 1) {
        return wave_reduce16(v, lane);
    } else {
        const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
        float b[4];
        if constexpr (RPI == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float send = b4 ? v[i] : v[i + 4];
                float keep = b4 ? v[i + 4] : v[i];
                b[i] = keep + __shfl_xor(send, 16, kWave);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = v[i];
        }
        float c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float send = b3 ? b[i] : b[i + 2];
            float keep = b3 ? b[i + 2] : b[i];
            c[i] = keep + __shfl_xor(send, 8, kWave);
        }
        float send = b2 ? c[0] : c[1];
        float keep = b2 ? c[1] : c[0];
        float d = keep + __shfl_xor(send, 4, kWave);
        d += __shfl_xor(d, 2, kWave);
        d += __shfl_xor(d, 1, kWave);
        return d;
    }
}
// the probability of slot RPI * t + (lane group), from the lanes that hold it (p_lane as the reduction left it)
template <int RPI>
__device__ __forceinline__ float rpi_prob(float p_lane, int t, int lane) {
    const float p0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p_lane), rpi_lane_of_slot<RPI>(RPI * t)));
    if constexpr (RPI == 1) {
        return p0;
    } else if constexpr (RPI == 2) {
        const float p1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p_lane), rpi_lane_of_slot<RPI>(RPI * t + 1)));
        return (lane & 32) ? p1 : p0;
    } else {
        const float p1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p_lane), rpi_lane_of_slot<RPI>(RPI * t + 1)));
        const float p2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p_lane), rpi_lane_of_slot<RPI>(RPI * t + 2)));
        const float p3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p_lane), rpi_lane_of_slot<RPI>(RPI * t + 3)));
        const float lo = (lane & 16) ? p1 : p0, hi = (lane & 16) ? p3 : p2;
        return (lane & 32) ? hi : lo;
    }
}
// sum of a value over the RPI lane groups (lanes l, l + 64 / RPI, ...), in every lane
template <int RPI>
__device__ __forceinline__ float rpi_group_sum(float v) {
    if constexpr (RPI >= 2) v += __shfl_xor(v, 32, kWave);
    if constexpr (RPI >= 4) v += __shfl_xor(v, 16, kWave);
    return v;
}

}  // namespace mli
