// Standalone HBM read micro-benchmark: what can a pure streaming read reach on this box, for the
// access shapes the decode kernels use?  Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each wave reads `rows_per_wave` rows of `row_f4` float4 (contiguous), rows spaced `stride_f4` apart.
template <int UNROLL, bool NT = false>
__global__ __launch_bounds__(256) void read_rows(const float4* __restrict__ src, float* __restrict__ sink,
                                                  long rows_total, int row_f4, long stride_f4) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    const int per_row = (row_f4 + 63) / 64;
    for (long r0 = wave * UNROLL; r0 < rows_total; r0 += nwaves * UNROLL) {
        for (int j = 0; j < per_row; ++j) {
            float4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                long r = r0 + u;
                int i = lane + j * 64;
                if (r < rows_total && i < row_f4) {
                    if (NT) {
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        f4v t = __builtin_nontemporal_load((const f4v __attribute__((address_space(1)))*)(src + r * stride_f4 + i));
                        v[u] = make_float4(t.x, t.y, t.z, t.w);
                    } else {
                        v[u] = src[r * stride_f4 + i];
                    }
                } else {
                    v[u] = make_float4(0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const f32x4_t __attribute__((address_space(1)))* gv4_ptr;

// each wave streams its own contiguous region of `bytes_per_wave`, UNROLL x 1 KiB in flight, optional nt hint
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_regions(const float4* __restrict__ src, float* __restrict__ sink,
                                                     long n4, long f4_per_wave) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (long base = wave * f4_per_wave; base + f4_per_wave <= n4; base += nwaves * f4_per_wave) {
        for (long o = 0; o < f4_per_wave; o += 64 * UNROLL) {
            f32x4_t v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                gv4_ptr p = (gv4_ptr)(src + base + o + u * 64 + lane);
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// the decode scan's order inside a page: K rows 0-7, K rows 8-15, V rows 0-7, V rows 8-15 (1 KiB pieces, bf16 D=512),
// UNROLL = 8 rows per batch, one batch consumed while the next is in flight is NOT modelled (plain issue / wait)
template <bool NT>
__global__ __launch_bounds__(256) void read_blocks_kv_order(const float4* __restrict__ src, float* __restrict__ sink,
                                                             const int* __restrict__ perm, int nblocks, long stride_f4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (long bi = wave; bi < nblocks; bi += nwaves) {
        const long base = (long)perm[bi] * 16 * stride_f4;
#pragma unroll
        for (int phase = 0; phase < 4; ++phase) {
            const int seg = phase < 2 ? 1 : 2;           // K, K, V, V
            const int r0 = (phase & 1) * 8;
            f4v v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f4v __attribute__((address_space(1)))* p =
                    (const f4v __attribute__((address_space(1)))*)(src + base + (long)(r0 + u) * stride_f4 + seg * 64 + lane);
                v[u] = NT ? __builtin_nontemporal_load(p) : *p;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// the decode scan's WORKGROUP structure around the same loads: an item = 32 blocks for the 4 waves of a workgroup
// (8 each, K 0-7, K 8-15, V 0-7, V 8-15 per block), with what the scan does around an item: block indices staged in
// LDS behind a barrier (the page pointers), and at the end the waves meet at a barrier, combine 2 KiB through LDS and
// store it (the chunk's partial result).  ITEM_SYNC = false drops the barriers and the LDS / store epilogue.
template <bool NT, bool ITEM_SYNC>
__global__ __launch_bounds__(256) void read_blocks_items(const float4* __restrict__ src, float* __restrict__ sink,
                                                          const int* __restrict__ perm, int nblocks, long stride_f4,
                                                          float* __restrict__ partial) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    __shared__ int idx_sh[32];
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    const int nitems = nblocks / 32;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        if (ITEM_SYNC) {
            if (threadIdx.x < 32) idx_sh[threadIdx.x] = perm[item * 32 + threadIdx.x];
            __syncthreads();
        }
        for (int k = wave; k < 32; k += 4) {
            const int blk = ITEM_SYNC ? __builtin_amdgcn_readfirstlane(idx_sh[k]) : perm[item * 32 + k];
            const long base = (long)blk * 16 * stride_f4;
#pragma unroll
            for (int phase = 0; phase < 4; ++phase) {
                const int seg = phase < 2 ? 1 : 2;
                const int r0 = (phase & 1) * 8;
                f4v v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const f4v __attribute__((address_space(1)))* p =
                        (const f4v __attribute__((address_space(1)))*)(src + base + (long)(r0 + u) * stride_f4 + seg * 64 + lane);
                    v[u] = NT ? __builtin_nontemporal_load(p) : *p;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
            }
        }
        if (ITEM_SYNC) {
            for (int i = lane; i < 512; i += 64) red[wave][i] = acc + i;
            __syncthreads();
            for (int i = threadIdx.x; i < 512; i += 256)
                partial[(long)item * 512 + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
            __syncthreads();
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// page-shaped access: 16 rows per block, blocks visited through a permutation (scattered pages); UNROLL rows of
// one block in flight per wave; reads `row_f4` float4 at `col_f4` of every row (K only, or K|V)
template <int UNROLL, bool NT, int EXTRA = 0>
__global__ __launch_bounds__(256) void read_blocks(const float4* __restrict__ src, float* __restrict__ sink,
                                                    const int* __restrict__ perm, int nblocks, int row_f4, int col_f4,
                                                    long stride_f4) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    const int per_row = (row_f4 + 63) / 64;
    for (long bi = wave; bi < nblocks; bi += nwaves) {
        const long base = (long)perm[bi] * 16 * stride_f4 + col_f4;
        for (int r0 = 0; r0 < 16; r0 += UNROLL) {
            for (int j = 0; j < per_row; ++j) {
                f4v v[UNROLL];
                const int i = lane + j * 64;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const f4v z = {0, 0, 0, 0};
                    if (i < row_f4) {
                        const f4v __attribute__((address_space(1)))* p = (const f4v __attribute__((address_space(1)))*)(src + base + (long)(r0 + u) * stride_f4 + i);
                        v[u] = NT ? __builtin_nontemporal_load(p) : *p;
                    } else {
                        v[u] = z;
                    }
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    acc += v[u].x + v[u].y + v[u].z + v[u].w;
                    // EXTRA independent-ish VALU ops per 16-byte load (4 accumulators), to mimic a kernel's arithmetic
                    float a0 = v[u].x, a1 = v[u].y, a2 = v[u].z, a3 = v[u].w;
#pragma unroll
                    for (int k = 0; k < EXTRA / 4; ++k) {
                        a0 = fmaf(a0, 1.0001f, a1); a1 = fmaf(a1, 0.9999f, a2); a2 = fmaf(a2, 1.0002f, a3); a3 = fmaf(a3, 0.9998f, a0);
                    }
                    acc += a0 + a1 + a2 + a3;
                }
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void copy_f4(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x);
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

template <typename F>
float time_ms(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

// MEMBENCH_LDS=<bytes of dynamic LDS per workgroup>: limits the workgroups per CU (70000 -> 2 per CU = the occupancy of
// the decode scan kernel) without touching the kernels
static unsigned g_lds = 0;

int main() {
    if (const char* e = getenv("MEMBENCH_LDS")) g_lds = (unsigned)atoi(e);
    std::printf("dynamic LDS per workgroup: %u bytes\n", g_lds);
    const long n4 = 1L << 28;  // 4 GiB source
    float4 *src, *dst; float* sink;
    CK(hipMalloc(&src, n4 * 16)); CK(hipMalloc(&dst, n4 * 16)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, n4 * 16));
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_f4<4>, dim3(grid), dim3(256), g_lds, 0, src, dst, n4); }, 5);
        printf("copy unroll4 grid %5d: %.1f GB/s (r+w)\n", grid, 2.0 * n4 * 16 / ms / 1e6);
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_f4<8>, dim3(4096), dim3(256), g_lds, 0, src, dst, n4); }, 5);
        printf("copy unroll8 grid  4096: %.1f GB/s (r+w)\n", 2.0 * n4 * 16 / ms / 1e6);
        ms = time_ms([&] { CK(hipMemcpyAsync(dst, src, n4 * 16, hipMemcpyDeviceToDevice, 0)); }, 5);
        printf("hipMemcpy D2D         : %.1f GB/s (r+w)\n", 2.0 * n4 * 16 / ms / 1e6);
    }
    for (long kib : {16, 64, 256}) {
        const long f4w = kib * 64;                      // float4 per wave region
        if (f4w % (64 * 16) != 0 || f4w > n4) continue;  // a region must hold whole UNROLL=16 batches (bounds!)
        for (int grid : {2048, 8192}) {
            float a = time_ms([&] { hipLaunchKernelGGL((read_regions<4, false>), dim3(grid), dim3(256), g_lds, 0, src, sink, n4, f4w); }, 5);
            float b = time_ms([&] { hipLaunchKernelGGL((read_regions<4, true>), dim3(grid), dim3(256), g_lds, 0, src, sink, n4, f4w); }, 5);
            float c = time_ms([&] { hipLaunchKernelGGL((read_regions<16, true>), dim3(grid), dim3(256), g_lds, 0, src, sink, n4, f4w); }, 5);
            printf("read regions %4ld KiB/wave grid %5d: u4 %.0f  u4-nt %.0f  u16-nt %.0f GB/s\n", kib, grid,
                   n4 * 16.0 / a / 1e6, n4 * 16.0 / b / 1e6, n4 * 16.0 / c / 1e6);
        }
    }
    // contiguous rows of 2 KiB (128 float4): pure streaming read
    struct Shape { const char* name; int row_f4; long stride_f4; };
    Shape shapes[] = {{"contig 2KiB rows", 128, 128}, {"2KiB @ 6KiB stride (K of fp32 pages D=512)", 128, 384},
                      {"4KiB @ 6KiB stride (K|V of fp32 pages)", 256, 384}, {"1KiB @ 3KiB stride (K of bf16 pages)", 64, 192},
                      {"2KiB @ 3KiB stride (K|V of bf16 pages)", 128, 192}};
    for (auto& sh : shapes) {
        long rows = (n4 - sh.row_f4) / sh.stride_f4;
        for (int grid : {2048, 4096, 8192}) {
            float ms4 = time_ms([&] { hipLaunchKernelGGL(read_rows<4>, dim3(grid), dim3(256), g_lds, 0, src, sink, rows, sh.row_f4, sh.stride_f4); }, 5);
            float ms8 = time_ms([&] { hipLaunchKernelGGL(read_rows<8>, dim3(grid), dim3(256), g_lds, 0, src, sink, rows, sh.row_f4, sh.stride_f4); }, 5);
            float ms16 = time_ms([&] { hipLaunchKernelGGL(read_rows<16>, dim3(grid), dim3(256), g_lds, 0, src, sink, rows, sh.row_f4, sh.stride_f4); }, 5);
            float nt8 = time_ms([&] { hipLaunchKernelGGL((read_rows<8, true>), dim3(grid), dim3(256), g_lds, 0, src, sink, rows, sh.row_f4, sh.stride_f4); }, 5);
            float nt16 = time_ms([&] { hipLaunchKernelGGL((read_rows<16, true>), dim3(grid), dim3(256), g_lds, 0, src, sink, rows, sh.row_f4, sh.stride_f4); }, 5);
            double bytes = (double)rows * sh.row_f4 * 16;
            printf("read %-44s grid %5d: u4 %.0f  u8 %.0f  u16 %.0f | nt u8 %.0f  nt u16 %.0f GB/s\n", sh.name, grid, bytes / ms4 / 1e6, bytes / ms8 / 1e6, bytes / ms16 / 1e6, bytes / nt8 / 1e6, bytes / nt16 / 1e6);
        }
    }
    {   // scattered page blocks: 16 token slots per block, slot = stride_f4 float4 (x | K | V)
        struct PShape { const char* name; int row_f4, col_f4; long stride_f4; };
        PShape ps[] = {{"bf16 D=512: K only   (1 KiB of every 3 KiB, 48 KiB blocks)", 64, 64, 192},
                       {"bf16 D=512: K|V      (2 KiB of every 3 KiB, 48 KiB blocks)", 128, 64, 192},
                       {"fp32 D=512: K|V      (4 KiB of every 6 KiB, 96 KiB blocks)", 256, 128, 384}};
        for (auto& sh : ps) {
            const int nblocks = (int)(n4 / (16 * sh.stride_f4));  // every block lies inside the 4 GiB buffer
            std::vector<int> lin(nblocks), shuf(nblocks);
            for (int i = 0; i < nblocks; ++i) lin[i] = shuf[i] = i;
            unsigned long long seed = 88172645463325252ULL;
            for (int i = nblocks - 1; i > 0; --i) {  // Fisher-Yates with xorshift
                seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
                int j = (int)(seed % (unsigned long long)(i + 1));
                int t = shuf[i]; shuf[i] = shuf[j]; shuf[j] = t;
            }
            int* dperm;
            CK(hipMalloc(&dperm, sizeof(int) * nblocks));
            double bytes = (double)nblocks * 16 * sh.row_f4 * 16;
            for (int mode = 0; mode < 2; ++mode) {
                CK(hipMemcpy(dperm, mode ? shuf.data() : lin.data(), sizeof(int) * nblocks, hipMemcpyHostToDevice));
                for (int grid : {2048, 8192}) {
                    float a = time_ms([&] { hipLaunchKernelGGL((read_blocks<8, true>), dim3(grid), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.row_f4, sh.col_f4, sh.stride_f4); }, 5);
                    float b = time_ms([&] { hipLaunchKernelGGL((read_blocks<16, true>), dim3(grid), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.row_f4, sh.col_f4, sh.stride_f4); }, 5);
                    float c8 = time_ms([&] { hipLaunchKernelGGL((read_blocks<8, true, 8>), dim3(grid), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.row_f4, sh.col_f4, sh.stride_f4); }, 5);
                    float c16 = time_ms([&] { hipLaunchKernelGGL((read_blocks<8, true, 16>), dim3(grid), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.row_f4, sh.col_f4, sh.stride_f4); }, 5);
                    float c32 = time_ms([&] { hipLaunchKernelGGL((read_blocks<8, true, 32>), dim3(grid), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.row_f4, sh.col_f4, sh.stride_f4); }, 5);
                    printf("blocks %-62s %s grid %5d: nt u8 %.0f  nt u16 %.0f | u8 + 8/16/32 VALU per load: %.0f %.0f %.0f GB/s\n", sh.name, mode ? "shuffled" : "linear  ", grid, bytes / a / 1e6, bytes / b / 1e6, bytes / c8 / 1e6, bytes / c16 / 1e6, bytes / c32 / 1e6);
                    if (sh.row_f4 == 128 && sh.stride_f4 == 192 && grid == 8192) {
                        float* part;
                        CK(hipMalloc(&part, sizeof(float) * 512 * (size_t)(nblocks / 32 + 1)));
                        for (int g2 : {512, 2048}) {
                            float y = time_ms([&] { hipLaunchKernelGGL((read_blocks_items<true, true>), dim3(g2), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.stride_f4, part); }, 5);
                            float n = time_ms([&] { hipLaunchKernelGGL((read_blocks_items<true, false>), dim3(g2), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.stride_f4, part); }, 5);
                            printf("       workgroup items of 32 blocks, grid %4d: with barriers + LDS merge + partial store %.0f GB/s, without %.0f GB/s\n", g2, (double)(nblocks / 32) * 32 * 16 * sh.row_f4 * 16 / y / 1e6, (double)(nblocks / 32) * 32 * 16 * sh.row_f4 * 16 / n / 1e6);
                        }
                        CK(hipFree(part));
                    }
                    if (sh.row_f4 == 128 && sh.stride_f4 == 192) {
                        float k = time_ms([&] { hipLaunchKernelGGL((read_blocks_kv_order<true>), dim3(grid), dim3(256), g_lds, 0, src, sink, dperm, nblocks, sh.stride_f4); }, 5);
                        printf("       same blocks in the scan's order (K 0-7, K 8-15, V 0-7, V 8-15): %.0f GB/s\n", bytes / k / 1e6);
                    }
                }
            }
            CK(hipFree(dperm));
        }
    }
    return 0;
}
