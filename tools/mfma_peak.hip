// Practical MFMA ceilings of the box (dev tool): waves doing nothing but v_mfma on register operands.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip && ./tools/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS>
__global__ __launch_bounds__(256) void f32_kernel(float* out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void bf16_kernel(float* out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(i * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// what one wave per SIMD reaches with 6 independent accumulators (the decode projection's MFMA waves), and what a
// clock64() tick is worth: ticks per MFMA and ticks per wall-clock nanosecond (wall_clock64() counts at 100 MHz)
__global__ __launch_bounds__(256) void bf16_six_kernel(float* out, unsigned long long* stamps, int iters) {
    f32x16 acc[6];
    for (int c = 0; c < 6; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f); b[i] = (__bf16)(i * 0.5f); }
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < 6; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = w1 - w0;
    }
}

// the same with operands that look like data: two fragment sets of random bf16 (uniform in [-1, 1)), a[2] x b[3] per set as
// in the decode projection's k sub-step, the sets alternating -- what the matrix pipe sustains when its inputs toggle
__global__ __launch_bounds__(256) void bf16_six_random_kernel(float* out, unsigned long long* stamps, const bf16x8* frags, int iters) {
    f32x16 acc[6];
    for (int c = 0; c < 6; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    bf16x8 a[2][2], b[2][3];
    const bf16x8* f = frags + (size_t)(blockIdx.x * 256 + threadIdx.x) * 10;
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < 2; ++i) a[s][i] = f[s * 5 + i];
        for (int i = 0; i < 3; ++i) b[s][i] = f[s * 5 + 2 + i];
    }
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[mt * 3 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][mt], b[s][nt], acc[mt * 3 + nt], 0, 0, 0);
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < 6; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = w1 - w0;
    }
}

template <class F>
static double time_ms(F&& launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    const int iters = 4096;
    for (int wgs_per_cu : {1, 2, 4}) {
        const int grid = 256 * wgs_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(f32_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters); }, 5);
        double flop = (double)grid * 4 * iters * 2 * (2.0 * 32 * 32 * 2);
        std::printf("fp32 32x32x2, 2 chains, %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(bf16_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters); }, 5);
        flop = (double)grid * 4 * iters * 2 * (2.0 * 32 * 32 * 16);
        std::printf("bf16 32x32x16, 2 chains, %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
    }
    for (int wgs_per_cu : {1, 2, 4, 6}) {
        const int grid = 256 * wgs_per_cu;
        double ms = time_ms([&] { hipLaunchKernelGGL(f32_kernel<1>, dim3(grid), dim3(256), 0, 0, out, 16 * iters); }, 3);
        double flop = (double)grid * 4 * 16 * iters * 1 * (2.0 * 32 * 32 * 2);
        std::printf("fp32 32x32x2, 1 dependent chain, %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wgs_per_cu, ms, flop / ms / 1e9);
    }
    {
        unsigned long long* stamps;
        hipMalloc(&stamps, 512 * sizeof(unsigned long long));
        for (int it : {768, 6144}) {   // 768 iterations x 6 = the decode projection's MFMA count per wave at D = 2048
            double ms = time_ms([&] { hipLaunchKernelGGL(bf16_six_kernel, dim3(256), dim3(256), 0, 0, out, stamps, it); }, 20);
            unsigned long long h[512];
            hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
            double ticks = 0, wall = 0;
            for (int i = 0; i < 256; ++i) { ticks += h[2 * i]; wall += h[2 * i + 1]; }
            ticks /= 256; wall /= 256;
            std::printf("bf16 32x32x16, 6 chains, 1 wave per SIMD, %d MFMAs per wave: %.4f ms  %.1f TFLOP/s; clock64 ticks per MFMA %.2f, "
                        "ticks per ns %.3f (in-kernel span %.2f us)\n", it * 6, ms, 256.0 * 4 * it * 6 * 32768.0 / ms / 1e9,
                        ticks / (it * 6.0), ticks / (wall * 10.0), wall / 100.0);
        }
        {
            const size_t n = (size_t)256 * 256 * 10 * 8;
            unsigned short* h = new unsigned short[n];
            unsigned long long x = 88172645463325252ull;
            for (size_t i = 0; i < n; ++i) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                const float v = (float)((x >> 11) & 0xffffff) / 8388608.f - 1.f;   // [-1, 1)
                unsigned int u; __builtin_memcpy(&u, &v, 4);
                h[i] = (unsigned short)(u >> 16);
            }
            bf16x8* frags;
            hipMalloc(&frags, n * 2);
            hipMemcpy(frags, h, n * 2, hipMemcpyHostToDevice);
            for (int it : {768, 6144}) {
                double ms = time_ms([&] { hipLaunchKernelGGL(bf16_six_random_kernel, dim3(256), dim3(256), 0, 0, out, stamps, frags, it); }, 50);
                unsigned long long hs[512];
                hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
                double ticks = 0, wall = 0;
                for (int i = 0; i < 256; ++i) { ticks += hs[2 * i]; wall += hs[2 * i + 1]; }
                ticks /= 256; wall /= 256;
                std::printf("bf16 32x32x16, random operands, 1 wave per SIMD, %d MFMAs per wave: %.4f ms  %.1f TFLOP/s; clock64 ticks per MFMA "
                            "%.2f, ticks per ns %.3f (in-kernel span %.2f us)\n", it * 6, ms, 256.0 * 4 * it * 6 * 32768.0 / ms / 1e9,
                            ticks / (it * 6.0), ticks / (wall * 10.0), wall / 100.0);
            }
            hipFree(frags);
            delete[] h;
        }
        hipFree(stamps);
    }
    // long run: does the rate hold (clocks under sustained MFMA load)?
    double ms = time_ms([&] { hipLaunchKernelGGL(f32_kernel<2>, dim3(1024), dim3(256), 0, 0, out, 65536); }, 3);
    std::printf("fp32 sustained (%.1f ms per launch): %.1f TFLOP/s\n", ms, 1024.0 * 4 * 65536 * 2 * 4096 / ms / 1e9);
    hipFree(out);
    return 0;
}
