// SURVEY 8(f) row 4 -- "page-pool layout v2": would splitting K | V from the input-embedding segment make the decode scan
// faster?  The reference's block is 16 token slots of [x | K | V] (include/utils.h:37,43): the scan reads 2/3 of every
// block, in pieces of one K or V row (1-2 KiB) at a stride of three rows.  Layout v2 would hold [K | V] only: every
// block a contiguous 2 * 16 * D * e bytes.  This probe reads the SAME number of blocks, visited through the SAME
// shuffled permutation, in the scan's order (K rows 0-7, K rows 8-15, V rows 0-7, V rows 8-15; 8 rows in flight per wave,
// non-temporal 16-byte lane loads, 2 workgroups of 4 waves per CU looping over the blocks) from both layouts and prints
// one JSON object per shape.  Build: hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o tools/layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(1)))* g4;

// ROW_F4 = 16-byte lane loads per K (or V) row / 64 (1 = 1 KiB rows, 2 = 2 KiB rows)
template <int NJ>
__global__ __launch_bounds__(256, 2) void scan_order_read(const float4* __restrict__ pool, float* __restrict__ sink,
                                                           const int* __restrict__ perm, int nblocks, long block_f4,
                                                           long slot_f4, long k_off_f4, long v_off_f4) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (long bi = wave; bi < nblocks; bi += nwaves) {
        const float4* base = pool + (long)perm[bi] * block_f4;
#pragma unroll
        for (int phase = 0; phase < 4; ++phase) {
            const long seg = phase < 2 ? k_off_f4 : v_off_f4;
            const int r0 = (phase & 1) * 8;
            f4v v[8][NJ];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    v[u][j] = __builtin_nontemporal_load((g4)(base + (long)(r0 + u) * slot_f4 + seg + j * 64 + lane));
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc += v[u][j].x + v[u][j].y + v[u][j].z + v[u][j].w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <class F>
static float time_ms(F&& f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    struct Shape { const char* name; int row_bytes; };
    const Shape shapes[] = {{"bf16 D=512 (config 4)", 1024}, {"fp32 D=256 (configs 2/3)", 1024}, {"fp32 D=512", 2048}};
    float* sink;
    CK(hipMalloc(&sink, 4));
    printf("[\n");
    bool first = true;
    for (const Shape& sh : shapes) {
        const long row_f4 = sh.row_bytes / 16;
        const long v1_block_f4 = 16 * 3 * row_f4, v2_block_f4 = 16 * 2 * row_f4;
        const int nblocks = (int)((6L << 30) / (v1_block_f4 * 16));  // a 6 GiB pool in the reference layout
        float4 *p1, *p2;
        CK(hipMalloc(&p1, (size_t)nblocks * v1_block_f4 * 16));
        CK(hipMalloc(&p2, (size_t)nblocks * v2_block_f4 * 16));
        CK(hipMemset(p1, 1, (size_t)nblocks * v1_block_f4 * 16));
        CK(hipMemset(p2, 1, (size_t)nblocks * v2_block_f4 * 16));
        std::vector<int> perm(nblocks);
        for (int i = 0; i < nblocks; ++i) perm[i] = i;
        unsigned long long seed = 88172645463325252ULL;
        for (int i = nblocks - 1; i > 0; --i) {  // Fisher-Yates with xorshift: the same order for both layouts
            seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
            const int j = (int)(seed % (unsigned long long)(i + 1));
            const int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        int* dperm;
        CK(hipMalloc(&dperm, sizeof(int) * nblocks));
        CK(hipMemcpy(dperm, perm.data(), sizeof(int) * nblocks, hipMemcpyHostToDevice));
        const double bytes = (double)nblocks * 32 * sh.row_bytes;   // K and V rows of 16 slots
        const int grid = 512;                                       // 2 workgroups per CU, looping
        auto run = [&](const float4* pool, long block_f4, long slot_f4, long k_off, long v_off) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                const float ms = time_ms([&] {
                    if (row_f4 == 64) hipLaunchKernelGGL(scan_order_read<1>, dim3(grid), dim3(256), 0, 0, pool, sink, dperm, nblocks, block_f4, slot_f4, k_off, v_off);
                    else hipLaunchKernelGGL(scan_order_read<2>, dim3(grid), dim3(256), 0, 0, pool, sink, dperm, nblocks, block_f4, slot_f4, k_off, v_off);
                }, 10);
                if (ms < best) best = ms;
            }
            return best;
        };
        // v1: slot = [x | K | V];  v2a: slot = [K | V] (token-major);  v2b: block = [K rows 0..15 | V rows 0..15] (segment-major)
        const float t1 = run(p1, v1_block_f4, 3 * row_f4, row_f4, 2 * row_f4);
        const float t2a = run(p2, v2_block_f4, 2 * row_f4, 0, row_f4);
        const float t2b = run(p2, v2_block_f4, row_f4, 0, 16 * row_f4);
        printf("%s {\"shape\": \"%s\", \"blocks\": %d, \"kv_bytes_read\": %.0f, \"pool_order\": \"shuffled\",\n"
               "  \"v1_reference_layout_x_k_v\": {\"ms\": %.4f, \"GBps\": %.0f},\n"
               "  \"v2_k_v_token_major\": {\"ms\": %.4f, \"GBps\": %.0f, \"vs_v1\": %.4f},\n"
               "  \"v2_k_rows_then_v_rows\": {\"ms\": %.4f, \"GBps\": %.0f, \"vs_v1\": %.4f}}",
               first ? "" : ",\n", sh.name, nblocks, bytes, t1, bytes / t1 / 1e6, t2a, bytes / t2a / 1e6, t1 / t2a, t2b,
               bytes / t2b / 1e6, t1 / t2b);
        first = false;
        CK(hipFree(p1)); CK(hipFree(p2)); CK(hipFree(dperm));
    }
    printf("\n]\n");
    return 0;
}
