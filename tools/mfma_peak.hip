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
    // long run: does the rate hold (clocks under sustained MFMA load)?
    double ms = time_ms([&] { hipLaunchKernelGGL(f32_kernel<2>, dim3(1024), dim3(256), 0, 0, out, 65536); }, 3);
    std::printf("fp32 sustained (%.1f ms per launch): %.1f TFLOP/s\n", ms, 1024.0 * 4 * 65536 * 2 * 4096 / ms / 1e9);
    hipFree(out);
    return 0;
}
