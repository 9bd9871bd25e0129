// bf16 MFMA GEMM for the bf16 paged path (BASELINE config 4): x[rows, D] . [Wk | Wq | Wv], bf16 operands,
// fp32 accumulation (v_mfma_f32_32x32x16_bf16), K/V written back to the pages as bf16, q as fp32.
// Same row gather / scatter front and back ends as the fp32 kernel (gemm_common.hpp); this file only differs in
// the tile engine:
//   * A (activations, k contiguous): staged row-major in LDS, fragments by ds_read_b128 (80-byte row stride:
//     conflict-free);
//   * B (weights, stored [k][n] with n contiguous, as the reference keeps them): staged row-major as loaded --
//     no transposing stores -- and read with ds_read_b64_tr_b16, the gfx950 transposing LDS read, which hands
//     each lane 4 consecutive k of its column (192-byte row stride: the 4 rows of a block land on disjoint
//     quarters of the 64 banks);
//   * 64x64x32 tile, 2x2 waves of 32x32, register-staged prefetch of the next tile under the MFMAs -- the
//     latency-bound shapes (config 4: 1.6 GFLOP per launch) and the prefill;
//   * 128x64x64 tile (MT = 2, KB = 64: each wave two 32x32 sub-tiles sharing every weight fragment, 8 MFMAs per
//     barrier pair instead of 2) for the decode projection of a large batch at a large emb_dim, where the small
//     tile spends its time on staging and barriers (B=1024, D=2048: 451 TFLOP/s).
#include <atomic>
#include <cstdlib>

#include "gemm_common.hpp"

namespace mli {

constexpr int HM = 64, HN = 64;   // (the k extent of a staged tile is the kernel's KB parameter)
constexpr int kLdbBytes = HN * 2 + 64;   // 192
constexpr int kHThreads = 256;

int fill_compact(int n_new);  // proj_gemm.hip
int latest_compact(int n_batch, int k_dim);
bool gemm_use_tall_tiles(int64_t tall_workgroups);
int deep_k_tiles_enabled();

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// -DMLI_GEMM_TRACE: where the k loop of the first workgroups spends its cycles (tools/gemm_trace.py) -- never the product
#ifdef MLI_GEMM_TRACE
constexpr int kGemmTraceWgs = 1024;
__device__ unsigned long long mli_gemm_trace[kGemmTraceWgs * 8];
#define MLI_GT(var) const unsigned long long var = clock64()
#define MLI_GT_ADD(i, a, b) do { if (threadIdx.x == 0) gt_acc[i] += (b) - (a); } while (0)
#else
#define MLI_GT(var) do { } while (0)
#define MLI_GT_ADD(i, a, b) do { } while (0)
#endif

union Frag8 {
    bf16x8_t v;
    uint4 u;
    s16x4_t h[2];
};

// MT = 32-row sub-tiles per wave (rows per workgroup = 64 * MT), KB = k extent of a staged tile (32 or 64)
// FP8 = the pages hold OCP e4m3 bytes (MLI_ELEM_FP8): A rows are widened to bf16 on their way into LDS (exact), K / V rows
// leave as fp8 (round to nearest even, saturating); the weights are bf16 and everything between is the bf16 kernel
template <int MODE, int MT = 1, int KB = 32, bool FP8 = false>
__global__ __launch_bounds__(kHThreads) void gemm_bf16_mfma_kernel(GemmArgs g) {
    constexpr int HM = 64 * MT;              // shadows the namespace-level 64
    constexpr int HK = KB;
    constexpr int kLdaBytes = HK * 2 + 16;   // 80 / 144: ds_read_b128 fragment reads are conflict-free
    constexpr int AP = HM * HK / (kHThreads * 8);   // 16-byte A loads per thread and tile
    constexpr int BP = HK * HN / (kHThreads * 8);   // 16-byte B loads per thread and tile
    constexpr int kAThreadsPerRow = HK / 8;
    __shared__ __align__(16) unsigned char As[HM * kLdaBytes];
    __shared__ __align__(16) unsigned char Bs[HK * kLdbBytes];
    __shared__ const float* a_ptr[HM];
    __shared__ float* o_ptr[HM];
    constexpr bool kFillMode = MODE == kPagedFill;
    __shared__ const float* e_ptr[kFillMode ? HM : 1];  // embedding prologue: fp32 emb_table / wpe row of every A row
    __shared__ const float* p_ptr[kFillMode ? HM : 1];

    const int tiles_n = (g.N + HN - 1) / HN;
    const int wsel = blockIdx.x / tiles_n;
    const int n0 = (blockIdx.x % tiles_n) * HN;
    const int m0 = blockIdx.y * HM;
    const int z = blockIdx.z;
    const int out_id = g.out_id[wsel];
    const uint16_t* __restrict__ W = reinterpret_cast<const uint16_t*>(g.w[wsel]);

    constexpr bool kLatest = MODE == kPagedLatest;
    __shared__ FillIndex fill_index[1];
    const int tid = threadIdx.x;
    int fill_total = 0;
    if (g.compact) {
        fill_total = build_fill_index<kHThreads, kLatest>(g, fill_index[0]);
        if (m0 >= fill_total) return;  // workgroup-uniform: every lane leaves together
    } else if (MODE == kPagedFill && m0 >= g.lengths[g.new_batch_idx[z]]) {
        return;
    }
    const bool embed = kFillMode && g.emb_table != nullptr;
    if (tid < HM) {
        RowDesc r{nullptr, nullptr, nullptr, nullptr};
        if (g.compact) {
            if (m0 + tid < fill_total) {
                int zz, ss;
                fill_index_lookup(fill_index[0], kLatest ? g.B : g.n_new, m0 + tid, zz, ss);
                r = kLatest ? resolve_row<MODE, true, FP8>(g, zz, 0, out_id) : resolve_row<MODE, true, FP8>(g, ss, zz, out_id);
            }
        } else {
            r = resolve_row<MODE, true, FP8>(g, m0 + tid, z, out_id);
        }
        a_ptr[tid] = r.a;
        o_ptr[tid] = r.o;
        if (kFillMode) {
            e_ptr[tid] = r.e;
            p_ptr[tid] = r.p;
        }
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave >> 1) * 32 * MT;
    const int wn = (wave & 1) * 32;

    // staging coordinates: 16-byte loads, AP + BP per thread and tile
    const int a_row = tid / kAThreadsPerRow, a_k8 = (tid % kAThreadsPerRow) * 8;  // A: rows x chunks of 8 k
    constexpr int kARowsPerPass = kHThreads / kAThreadsPerRow;
    const int b_row = tid >> 3, b_n8 = (tid & 7) * 8;                            // B: 32 k-rows x 8 chunks of 8 n
    uint4 a_reg[AP], b_reg[BP];
    const uint16_t* a_src[AP];
    const float* e_src[AP];
    const float* p_src[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        a_src[p] = reinterpret_cast<const uint16_t*>(a_ptr[a_row + p * kARowsPerPass]);
        e_src[p] = kFillMode ? e_ptr[a_row + p * kARowsPerPass] : nullptr;
        p_src[p] = kFillMode ? p_ptr[a_row + p * kARowsPerPass] : nullptr;
    }
    const bool writes_x = blockIdx.x == 0;  // first column tile of the first weight: every A element passes once

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            a_reg[p] = make_uint4(0, 0, 0, 0);
            if (kFillMode && embed && a_src[p] != nullptr) {
                // encoder as prologue: x = bf16(emb[tok] + wpe[s]) -- fp32 sum, one rounding, what the bf16 encoder
                // kernel writes -- fed to the MFMA and, by the first column tile, written to the page's segment 0
                if (k0 + a_k8 < g.K) {
                    const float4 e0 = *reinterpret_cast<const float4*>(e_src[p] + k0 + a_k8);
                    const float4 e1 = *reinterpret_cast<const float4*>(e_src[p] + k0 + a_k8 + 4);
                    const float4 p0 = *reinterpret_cast<const float4*>(p_src[p] + k0 + a_k8);
                    const float4 p1 = *reinterpret_cast<const float4*>(p_src[p] + k0 + a_k8 + 4);
                    if constexpr (FP8) {   // x = fp8(emb[tok] + wpe[s]): fp32 sum, one rounding; the MFMA sees what the page holds
                        const uint2 x8 = make_uint2(f32x4_to_fp8x4(e0.x + p0.x, e0.y + p0.y, e0.z + p0.z, e0.w + p0.w),
                                                    f32x4_to_fp8x4(e1.x + p1.x, e1.y + p1.y, e1.z + p1.z, e1.w + p1.w));
                        const uint2 lo = fp8x4_to_bf16x4(x8.x), hi = fp8x4_to_bf16x4(x8.y);
                        a_reg[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        if (writes_x)
                            *reinterpret_cast<uint2*>(const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(a_src[p])) + k0 + a_k8) = x8;
                    } else {
                        uint4 x;
                        x.x = f32_to_bf16(e0.x + p0.x) | ((uint32_t)f32_to_bf16(e0.y + p0.y) << 16);
                        x.y = f32_to_bf16(e0.z + p0.z) | ((uint32_t)f32_to_bf16(e0.w + p0.w) << 16);
                        x.z = f32_to_bf16(e1.x + p1.x) | ((uint32_t)f32_to_bf16(e1.y + p1.y) << 16);
                        x.w = f32_to_bf16(e1.z + p1.z) | ((uint32_t)f32_to_bf16(e1.w + p1.w) << 16);
                        a_reg[p] = x;
                        if (writes_x) *reinterpret_cast<uint4*>(const_cast<uint16_t*>(a_src[p]) + k0 + a_k8) = x;
                    }
                }
            } else if (a_src[p] != nullptr && k0 + a_k8 < g.K) {
                if constexpr (FP8) {   // 8 fp8 (8 bytes) -> 8 bf16
                    const uint2 x8 = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(a_src[p]) + k0 + a_k8);
                    const uint2 lo = fp8x4_to_bf16x4(x8.x), hi = fp8x4_to_bf16x4(x8.y);
                    a_reg[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    a_reg[p] = *reinterpret_cast<const uint4*>(a_src[p] + k0 + a_k8);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            b_reg[p] = make_uint4(0, 0, 0, 0);
            const int k = k0 + b_row + p * 32, n = n0 + b_n8;
            if (k < g.K && n < g.N) b_reg[p] = *reinterpret_cast<const uint4*>(W + (int64_t)k * g.N + n);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < AP; ++p)
            *reinterpret_cast<uint4*>(&As[(a_row + p * kARowsPerPass) * kLdaBytes + a_k8 * 2]) = a_reg[p];
#pragma unroll
        for (int p = 0; p < BP; ++p)
            *reinterpret_cast<uint4*>(&Bs[(b_row + p * 32) * kLdbBytes + b_n8 * 2]) = b_reg[p];
    };

    f32x16_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    // fragment addresses (constant over the k loop)
    const int li = lane & 31, lh = lane >> 5;
    const unsigned a_off = (unsigned)((wm + li) * kLdaBytes + lh * 16);
    // transposing read: lane 4q+p of a 16-lane group points at row q, columns 4p..4p+3 of the group's 4x16 block
    const int grp_col = wn + 16 * ((lane >> 4) & 1);
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const unsigned b_off = (unsigned)((8 * lh + tq) * kLdbBytes + (grp_col + 4 * tp) * 2);

    const int nk = (g.K + HK - 1) / HK;
#ifdef MLI_GEMM_TRACE
    unsigned long long gt_acc[5] = {0, 0, 0, 0, 0};
    const unsigned long long gt_begin = clock64();
#endif
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        MLI_GT(g0);
        if (t + 1 < nk) load_tile((t + 1) * HK);
        MLI_GT(g1);
        MLI_GT_ADD(0, g0, g1);
#pragma unroll
        for (int kk = 0; kk < HK; kk += 16) {
            Frag8 b;
            b.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Bs[b_off + kk * kLdbBytes]));
            b.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Bs[b_off + (kk + 4) * kLdbBytes]));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                Frag8 a;
                a.u = *reinterpret_cast<const uint4*>(&As[a_off + mt * 32 * kLdaBytes + kk * 2]);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[mt], 0, 0, 0);
            }
        }
        MLI_GT(g2);
        MLI_GT_ADD(1, g1, g2);
        __syncthreads();
        MLI_GT(g3);
        MLI_GT_ADD(2, g2, g3);
        if (t + 1 < nk) {
            store_tile();
            MLI_GT(g4);
            MLI_GT_ADD(3, g3, g4);
            __syncthreads();
            MLI_GT(g5);
            MLI_GT_ADD(4, g4, g5);
        }
    }
#ifdef MLI_GEMM_TRACE
    {
        const unsigned wg = blockIdx.x + gridDim.x * blockIdx.y;
        if (threadIdx.x == 0 && wg < (unsigned)kGemmTraceWgs) {
            for (int i = 0; i < 5; ++i) mli_gemm_trace[wg * 8 + i] = gt_acc[i];
            mli_gemm_trace[wg * 8 + 5] = clock64() - gt_begin;
            mli_gemm_trace[wg * 8 + 6] = gt_begin;
            mli_gemm_trace[wg * 8 + 7] = (unsigned long long)nk;
        }
    }
#endif

    // epilogue: register r of lane l is (row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mi = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int n = n0 + wn + li;
            float* op = o_ptr[mi];
            if (op != nullptr && n < g.N) {
                if (out_id == 1) op[n] = acc[mt][r];                                         // q: fp32
                else if (FP8) reinterpret_cast<uint8_t*>(op)[n] = (uint8_t)f32_to_fp8(acc[mt][r]);   // K / V: fp8 page rows
                else reinterpret_cast<uint16_t*>(op)[n] = f32_to_bf16(acc[mt][r]);            // K / V: bf16 page rows
            }
        }
    }
}

// ---- the large decode projection (>= 1024 rows at emb_dim >= 1024): LDS-DMA loader waves + MFMA waves ------------------
// In the kernel above one wave does everything in turn -- issue the next tile's global loads (the issue itself takes the
// time the CU's vector-memory path needs for the bytes), wait for them, write them to LDS, read fragments, multiply -- and
// only other waves of the SIMD overlap any of it: cycle stamps (-DMLI_GEMM_TRACE) put 25 % of the k loop into load issue,
// 24 % into waiting + LDS writes, 27 % into fragment reads + MFMAs, and neither a 128 x 128 tile nor two or three tiles in
// flight in registers change the total (47-50 us at B=1024, D=2048 for all of them).  Round 2 split the two jobs over the
// waves of a 512-thread workgroup (waves 4-7 load through their registers and ds_write, waves 0-3 multiply: 39-40 us);
// that kernel's loaders still moved every byte through registers and a ds_write pass (the slowest way into LDS, competing
// with the MFMA waves' fragment reads), two padded buffers of 39 KB left room for one tile in flight, and 128 x 128 tiles
// over 1024 x 3 x 2048 outputs were 384 workgroups = 1.5 per CU.  This kernel replaced it in round 3:
//   * the [Wk | Wq | Wv] columns are one sequence of 64-column sub-tiles and a workgroup takes three of them (192
//     columns, possibly of two weights): 1024 rows -> 8 x 32 = 256 workgroups, one per CU, dealt so that an XCD's 32
//     workgroups share 4 column tiles x all row tiles (its L2 sees 3 MB of weights + the rows);
//   * waves 4-7 only issue global_load_lds_dwordx4 (1 KiB per instruction, no registers, no ds_write): 16 for the A tile
//     (8 gathered rows x 128 B each), 24 for the B tile (8 k-rows x 128 B of a sub-tile); the LDS image is lane-linear
//     per instruction, so the bank swizzle is applied to the SOURCE address: A [128 rows][8 chunks of 16 B] with chunk c
//     stored at c ^ ((row >> 1) & 7) (ds_read_b128 fragment reads conflict-free), B [3][64 k][128 B] with the two 64-byte
//     halves of a k-row swapped when k & 2 (the 4 k-rows of a ds_read_b64_tr_b16 group land on 4 different bank
//     quarters) -- unpadded, 40 KiB per stage, three stages: two tiles in flight while the third is multiplied;
//   * the loaders retire a tile with a counted vmcnt (the younger tile stays in flight) and a raw s_barrier; the MFMA
//     waves (2 x 2, 64 x 96 each: 6 accumulator tiles, 5 KiB of fragments per 6 MFMAs) keep three fragment sets and read
//     two k sub-steps ahead, across that barrier.
// Same 32x32x16 MFMA steps in the same k order as the tiled kernel: pages and q_output bit-identical (tested).
constexpr int kDmThreads = 512;
constexpr int kDmM = 128, kDmK = 64;   // (x 192 columns = three 64-column sub-tiles)
constexpr int kDmABytes = kDmM * kDmK * 2;         // 16384: [128][128 B]
constexpr int kDmSubBytes = kDmK * 64 * 2;         //  8192: one 64-column sub-tile, [64 k][128 B]
constexpr int kDmStageBytes = kDmABytes + 3 * kDmSubBytes;   // 40960
constexpr int kDmStages = 3;
constexpr int kDmLoadsPerWave = (kDmStageBytes / 1024) / 4;  // 10 DMA instructions per loader wave and tile
constexpr int kDmOutSubBytes = kDmM * 64 * 4;      // epilogue: one 64-column sub-tile of the output, fp32 at most
constexpr size_t kDmSmem = (size_t)kDmStages * kDmStageBytes + kDmM * sizeof(void*);
static_assert(3 * kDmOutSubBytes <= kDmStages * kDmStageBytes, "the output tile reuses the stages");

// -DMLI_DMA_TRACE: where a workgroup of the kernel below spends its cycles (tools/gemm_dma_trace.py) -- never the product
#ifdef MLI_DMA_TRACE
constexpr int kDmaTraceWgs = 1024;
__device__ unsigned long long mli_dma_trace[kDmaTraceWgs * 16];
#define MLI_DT(var) const unsigned long long var = clock64()
#define MLI_DT_PUT(i, v) do { if (blockIdx.x < (unsigned)kDmaTraceWgs) mli_dma_trace[blockIdx.x * 16 + (i)] = (v); } while (0)
#else
#define MLI_DT(var) do { } while (0)
#define MLI_DT_PUT(i, v) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) void* global_void_ptr;

__device__ __forceinline__ void dma16(const unsigned char* src, unsigned char* lds_dst) {
    // every lane's 16 bytes go to lds_dst (wave-uniform) + 16 * lane; counted on vmcnt
    __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)lds_dst, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(kDmThreads) void gemm_bf16_dma_kernel(GemmArgs g, int col_tiles, int row_tiles) {
    static_assert(MODE == kPagedLatest, "decode projection only");
    extern __shared__ __align__(16) unsigned char dm_smem[];
    const uint16_t** a_ptr = reinterpret_cast<const uint16_t**>(dm_smem + kDmStages * kDmStageBytes);  // [kDmM]

    // tile of this workgroup.  Workgroups go to the XCDs round-robin: XCD x takes the column tiles
    // [x * col_tiles / 8, (x + 1) * col_tiles / 8) of every row tile
    int ct, rt;
    if (col_tiles % 8 == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, cpx = col_tiles >> 3;
        rt = idx / cpx;
        ct = xcd * cpx + idx % cpx;
    } else {
        rt = blockIdx.x / col_tiles;
        ct = blockIdx.x % col_tiles;
    }
    (void)row_tiles;
    const int m0 = rt * kDmM;
    const int subs_per_w = g.N >> 6;   // 64-column sub-tiles per weight matrix
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = g.K / kDmK;
    MLI_DT(dt_entry);
#ifdef MLI_DMA_TRACE
    const unsigned long long dt_wall0 = wall_clock64();   // 100 MHz: what a clock64() tick is worth under this load
#endif

    if (wave >= 4) {
        // ---------------- loader waves: 4 A + 6 B instructions per tile each ----------------
        const int lw = wave - 4;
        const int r8 = lane >> 3, c8 = lane & 7;
        const unsigned char* src[kDmLoadsPerWave];   // [0, 6): B, [6, 10): A
        int dst[kDmLoadsPerWave];
        const int64_t b_step = (int64_t)kDmK * g.N * 2;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = lw * 6 + i;
            const int sub = idx >> 3, kb = (idx & 7) * 8;
            const int k = kb + r8;
            const int sg = ct * 3 + sub;
            const unsigned char* W = reinterpret_cast<const unsigned char*>(g.w[sg / subs_per_w]);
            src[i] = W + ((int64_t)k * g.N + (sg % subs_per_w) * 64) * 2 + ((c8 ^ (((k >> 1) & 1) << 2)) << 4);
            dst[i] = kDmABytes + sub * kDmSubBytes + kb * 128;
        }
        auto issue_b = [&](int t) {
            unsigned char* stage = dm_smem + (t % kDmStages) * kDmStageBytes;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                dma16(src[i], stage + dst[i]);
                src[i] += b_step;
            }
        };
        auto issue_a = [&](int t) {
            unsigned char* stage = dm_smem + (t % kDmStages) * kDmStageBytes;
#pragma unroll
            for (int i = 6; i < kDmLoadsPerWave; ++i) {
                dma16(src[i], stage + dst[i]);
                src[i] += kDmK * 2;
            }
        };
        // the weights of the first two tiles travel while waves 0-1 look the rows up (raw barrier: a __syncthreads() would
        // wait for them)
        issue_b(0);
        if (nk > 1) issue_b(1);
        __builtin_amdgcn_s_barrier();   // the row pointers are in LDS
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * lw + 8 * i + r8;
            const uint16_t* p = a_ptr[row];
            // an empty slot's tile row is never stored: any readable bytes will do (a masked lane would change the
            // number of DMA instructions in flight, which the counted waits below rely on)
            if (p == nullptr) p = reinterpret_cast<const uint16_t*>(g.w[0]);
            src[6 + i] = reinterpret_cast<const unsigned char*>(p) + ((c8 ^ ((row >> 1) & 7)) << 4);
            dst[6 + i] = (32 * lw + 8 * i) * 128;
        }
        issue_a(0);
        if (nk > 1) issue_a(1);
#ifdef MLI_DMA_TRACE
        unsigned long long dt_wait = 0, dt_bar = 0, dt_issue = 0;
#endif
        for (int t = 0; t < nk; ++t) {
            // tile t has landed (this wave's part); tile t + 1 may stay in flight.  In issue order the queue holds
            // B0 B1 A0 A1 before the loop (t = 0: everything but A1 must be done) and [tile t: 10][tile t + 1: 10] afterwards
            MLI_DT(l0);
#ifdef MLI_DMA_TRACE
            if (g.n_new & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            if (t + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (t == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmLoadsPerWave) : "memory");
            MLI_DT(l1);
            __builtin_amdgcn_s_barrier();   // #t: tile t readable; every MFMA wave is done with tile t - 1
            asm volatile("" ::: "memory");
            MLI_DT(l2);
#ifdef MLI_DMA_TRACE
            if (g.n_new & 1) continue;   // debug: nothing is loaded after the first two tiles (the MFMA waves multiply stale bytes)
#endif
            if (t + 2 < nk) {   // into the stage tile t - 1 occupied
                issue_b(t + 2);
                issue_a(t + 2);
            }
#ifdef MLI_DMA_TRACE
            const unsigned long long l3 = clock64();
            dt_wait += l1 - l0; dt_bar += l2 - l1; dt_issue += l3 - l2;
#endif
        }
#ifdef MLI_DMA_TRACE
        if (tid == 256) { MLI_DT_PUT(8, dt_wait); MLI_DT_PUT(9, dt_bar); MLI_DT_PUT(10, dt_issue); }
#endif
        __builtin_amdgcn_s_barrier();   // #nk: the MFMA waves' barrier in front of the last tile's (unused) look-ahead
        __builtin_amdgcn_s_barrier();   // E1: (the MFMA waves are done with the last tile)
        __builtin_amdgcn_s_barrier();   // E2: the output tile is in LDS
        asm volatile("" ::: "memory");
    } else {
        // ---------------- MFMA waves: 2 x 2, 64 x 96 each ----------------
        if (tid < kDmM) a_ptr[tid] = reinterpret_cast<const uint16_t*>(resolve_row<MODE, true>(g, m0 + tid, 0, 0).a);
        __syncthreads();   // (pairs with the loaders' first barrier)
        MLI_DT(dt_rows);
        const int wm = (wave >> 1) * 64;
        const int wn = (wave & 1) * 96;
        f32x16_t acc[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        const int li = lane & 31, lh = lane >> 5;
        const int a_sw = (li >> 1) & 7;                       // (row >> 1) & 7: wm and mt * 32 are multiples of 16
        const unsigned a_row_off = (unsigned)((wm + li) * 128);
        const int tq = (lane >> 2) & 3, tp = lane & 3, g16 = (lane >> 4) & 1;
        unsigned b_off[3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = wn + nt * 32;
            b_off[nt] = (unsigned)(kDmABytes + (col >> 6) * kDmSubBytes + (8 * lh + tq) * 128 +
                                   ((((col >> 5) & 1) * 64 + g16 * 32 + tp * 8) ^ (((tq >> 1) & 1) << 6)));
        }
        // in the order the MFMAs want them: (a0, b0), b1, b2, a1
        auto read_frags = [&](const unsigned char* st, int q, Frag8 (&a)[2], Frag8 (&b)[3]) {
            const unsigned ac = (unsigned)(((2 * q + lh) ^ a_sw) << 4);
            a[0].u = *reinterpret_cast<const uint4*>(&st[a_row_off + ac]);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                b[nt].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&st[b_off[nt] + q * 16 * 128]));
                b[nt].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&st[b_off[nt] + (q * 16 + 4) * 128]));
            }
            a[1].u = *reinterpret_cast<const uint4*>(&st[a_row_off + 32 * 128 + ac]);
        };
        auto multiply = [&](const Frag8 (&a)[2], const Frag8 (&b)[3]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt].v, b[nt].v, acc[mt][nt], 0, 0, 0);
        };
        // one k sub-step: the 8 fragment reads of the NEXT sub-step go out between the first MFMAs of this one (2 per MFMA),
        // so that they have the rest of the sub-step to land; left alone the compiler issues them right in front of their use
        auto interleave = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 LDS reads
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        };
        // THREE fragment sets: the reads of k sub-step s + 2 go out while sub-step s is multiplied, so a fragment has a whole
        // sub-step (6 MFMAs) more to land than it takes the LDS under load (with two sets every sub-step began with a wait).  A
        // tile has 4 sub-steps, so set and stage indices repeat every 3 tiles: the loop body is 3 tiles with everything static.
        // Barrier #(t + 1) sits in front of sub-step 2 of tile t, whose look-ahead is the next tile's first fragments; every read
        // of tile t has been issued (and is waited for by the barrier's fence) by then, so the loaders may refill its stage.
        Frag8 fa[3][2], fb[3][3];
        __syncthreads();  // #0
        MLI_DT(dt_first);
#ifdef MLI_DMA_TRACE
        unsigned long long dt_mbar = 0;
#endif
        read_frags(dm_smem, 0, fa[0], fb[0]);
        read_frags(dm_smem, 1, fa[1], fb[1]);
        for (int t0 = 0; t0 < nk; t0 += 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = t0 + j;
                if (t < nk) {   // (workgroup-uniform)
                    const unsigned char* st = dm_smem + j * kDmStageBytes;               // t % 3 == j
                    const unsigned char* nx = dm_smem + ((j + 1) % 3) * kDmStageBytes;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cur = (4 * j + q) % 3, ahead = (4 * j + q + 2) % 3;
                        if (q == 2) {
                            MLI_DT(m0t);
                            __syncthreads();  // #(t + 1)
#ifdef MLI_DMA_TRACE
                            dt_mbar += clock64() - m0t;
#endif
                        }
                        // (no condition on the last tile: its look-ahead reads a stage nobody writes any more and is never
                        // multiplied -- a branch here would end the basic block, and with it the interleaving and hipcc's
                        // exact lgkmcnt bookkeeping, in every tile)
                        read_frags(q < 2 ? st : nx, (q + 2) & 3, fa[ahead], fb[ahead]);
                        multiply(fa[cur], fb[cur]);
                        interleave();
                    }
                }
            }
        }
        MLI_DT(dt_loop_end);
#ifdef MLI_DMA_TRACE
        if (tid == 0) {
            MLI_DT_PUT(0, dt_entry); MLI_DT_PUT(1, dt_rows); MLI_DT_PUT(2, dt_first); MLI_DT_PUT(3, dt_loop_end);
            MLI_DT_PUT(5, dt_mbar);
        }
#endif
        // the output tile goes to LDS as fp32 (no DMA is pending; E1: every wave is done with the last tile): per 64-column
        // sub-tile [128 rows][64 floats].  Register r of lane l is (row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31)
        __syncthreads();   // E1
#ifdef MLI_DMA_TRACE
        if (tid == 0) MLI_DT_PUT(11, clock64());
#endif
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int col = wn + nt * 32;
            unsigned char* C = dm_smem + (col >> 6) * kDmOutSubBytes + (((col >> 5) & 1) * 32 + li) * 4;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(&C[(wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 256]) = acc[mt][nt][r];
        }
        __syncthreads();   // E2
#ifdef MLI_DMA_TRACE
        if (tid == 0) MLI_DT_PUT(12, clock64());
#endif
    }

    // all 8 waves: whole rows of a sub-tile leave as 16-byte stores -- 256 contiguous bytes per q row (4 rows per
    // instruction), 128 per K / V row (8 rows per instruction).  Every piece is read into its own registers before the first
    // store goes out: a register that a store in flight still reads cannot be reloaded without waiting for that store
    const uint16_t* q_row[4];
    const uint16_t* kv_row[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) q_row[i] = a_ptr[wave * 16 + i * 4 + (lane >> 4)];
#pragma unroll
    for (int i = 0; i < 2; ++i) kv_row[i] = a_ptr[wave * 16 + i * 8 + (lane >> 3)];
    // (a wave-uniform choice per sub-tile.)  q: 4 rows per instruction, 16 lanes x 16 bytes each; K / V: a lane turns 8
    // floats into 8 bf16 -- 8 rows per instruction, 8 lanes x 16 bytes each
    auto load_sub = [&](int sub, uint4 (&v)[4]) -> int {
        const int out_id = g.out_id[(ct * 3 + sub) / subs_per_w];
        const unsigned char* C = dm_smem + sub * kDmOutSubBytes;
        if (out_id == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = *reinterpret_cast<const uint4*>(&C[(wave * 16 + i * 4 + (lane >> 4)) * 256 + (lane & 15) * 16]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = *reinterpret_cast<const uint4*>(&C[(wave * 16 + (i >> 1) * 8 + (lane >> 3)) * 256 + (lane & 7) * 32 + (i & 1) * 16]);
        }
        return out_id;
    };
    auto pack2 = [](uint32_t a, uint32_t b) { return (uint32_t)f32_to_bf16(__uint_as_float(a)) | ((uint32_t)f32_to_bf16(__uint_as_float(b)) << 16); };
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) u32x4_t* global_u32x4_ptr;
    auto store_sub = [&](int sub, int out_id, const uint4 (&v)[4]) {
        const int n = ((ct * 3 + sub) % subs_per_w) * 64;
        if (out_id == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 16 + i * 4 + (lane >> 4);
                if (q_row[i] != nullptr)
                    *(global_u32x4_ptr)(g.q_output + (int64_t)(m0 + row) * g.N + n + (lane & 15) * 4) = u32x4_t{v[i].x, v[i].y, v[i].z, v[i].w};
            }
        } else {
            const int64_t seg = (int64_t)(out_id == 0 ? kSegK : kSegV) * g.K;   // element offset from the token's x row
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u32x4_t w{pack2(v[2 * i].x, v[2 * i].y), pack2(v[2 * i].z, v[2 * i].w),
                                pack2(v[2 * i + 1].x, v[2 * i + 1].y), pack2(v[2 * i + 1].z, v[2 * i + 1].w)};
                if (kv_row[i] != nullptr) *(global_u32x4_ptr)(const_cast<uint16_t*>(kv_row[i]) + seg + n + (lane & 7) * 8) = w;
            }
        }
    };
    uint4 v0[4], v1[4], v2[4];
    const int o0 = load_sub(0, v0), o1 = load_sub(1, v1), o2 = load_sub(2, v2);
    store_sub(0, o0, v0);
    store_sub(1, o1, v1);
    store_sub(2, o2, v2);
#ifdef MLI_DMA_TRACE
    if (tid == 0) MLI_DT_PUT(13, clock64());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) { MLI_DT_PUT(4, clock64()); MLI_DT_PUT(6, wall_clock64() - dt_wall0); }
#endif
}

static thread_local int g_bf16_split = 1;  // mli_tune "gemm_bf16_split": the large decode projection runs 1 (default) = the
                                           // loader-wave / MFMA-wave kernel above, 0 = the 128 x 64 tiled kernel
void set_bf16_split(int v) { g_bf16_split = v != 0; }

int launch_latest_paged_bf16_native(uint16_t* const* page_table, const int* lengths, const uint16_t* wk,
                                    const uint16_t* wq, const uint16_t* wv, float* q, int B, int S, int D,
                                    hipStream_t st) {
    GemmArgs g{};
    g.w[0] = reinterpret_cast<const float*>(wk); g.w[1] = reinterpret_cast<const float*>(wq);
    g.w[2] = reinterpret_cast<const float*>(wv); g.n_out = 3;
    g.out_id[0] = 0; g.out_id[1] = 1; g.out_id[2] = 2;
    g.M = B; g.N = D; g.K = D;
    g.page_table = reinterpret_cast<float* const*>(page_table); g.q_output = q; g.lengths = lengths;
    g.B = B; g.S = S;
    g.compact = latest_compact(B, D);
    const int tiles_x = ceil_div_i(D, HN) * 3;
    // Where the loader-wave / MFMA-wave kernel runs (tools/gemm_bf16_shape_probe.py; "gemm_tall_tiles" = 0 keeps every launch on
    // the 64-row tiles): a workgroup takes ~20 us for its 128 x 192 tile whatever the grid, which beats the tiled kernels from
    // emb_dim 1536 at any batch (2048: 21 vs 34 us at 128 rows, 29 vs 53 us at 1024) and at emb_dim 1024 from ~320 rows
    // (14.3 vs 15.6 us at 512; 13.5 vs 12.0 us at 128: there the 64 x 64 tiles' 4x as many workgroups win)
    if (g_bf16_split && D % 64 == 0 && gemm_use_tall_tiles((int64_t)1 << 40) && (D >= 1536 || (D >= 1024 && B >= 320))) {
        static std::atomic<unsigned long long> dma_opted_in{0};  // > 64 KiB of dynamic LDS: opt in once per device
        int device = 0;
        (void)hipGetDevice(&device);
        const unsigned long long bit = 1ull << (device & 63);
        if (!(dma_opted_in.load(std::memory_order_relaxed) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dma_kernel<kPagedLatest>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDmSmem);
            if (e != hipSuccess) return (int)e;
            dma_opted_in.fetch_or(bit, std::memory_order_relaxed);
        }
        g.compact = 0;  // as below: an empty row's tile rows are never stored
#ifdef MLI_DMA_TRACE
        if (const char* dbg = std::getenv("MLI_DMA_DEBUG")) g.n_new = std::atoi(dbg);
#endif
        const int col_tiles = D / 64, row_tiles = ceil_div_i(B, kDmM);   // 3 weights x D / 192 columns
        hipLaunchKernelGGL((gemm_bf16_dma_kernel<kPagedLatest>), dim3(col_tiles * row_tiles), dim3(kDmThreads), kDmSmem, st,
                           g, col_tiles, row_tiles);
        return launch_status();
    }
    if (gemm_use_tall_tiles((int64_t)tiles_x * ceil_div_i(B, 128))) {  // as the fp32 kernel: >= 2 workgroups per CU
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedLatest, 2, 64>), dim3(tiles_x, ceil_div_i(B, 128), 1),
                           dim3(kHThreads), 0, st, g);
        return launch_status();
    }
    dim3 grid(tiles_x, ceil_div_i(B, HM), 1);
    // Short reductions are latency-bound (config 4: 16 k-steps of 32, each exposing a global-load round trip that
    // one workgroup per SIMD cannot hide): stage 128 k per tile instead -- 4 round trips.
    if (D >= 256 && D <= 1024 && deep_k_tiles_enabled())
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedLatest, 1, 128>), grid, dim3(kHThreads), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedLatest>), grid, dim3(kHThreads), 0, st, g);
    return launch_status();
}

int launch_fill_paged_bf16_embed(const float* emb_table, const float* wpe, const int* tokens,
                                 uint16_t* const* page_table, const int* new_idx, const int* lengths, const uint16_t* wk,
                                 const uint16_t* wv, int B, int S, int D, int n_new, hipStream_t st) {
    if (n_new == 0) return 0;
    if (n_new < 0 || B <= 0 || S % kPage != 0 || D % 8 != 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.emb_table = emb_table; g.wpe = wpe; g.inp = tokens;
    g.w[0] = reinterpret_cast<const float*>(wk); g.w[1] = reinterpret_cast<const float*>(wv); g.n_out = 2;
    g.out_id[0] = 0; g.out_id[1] = 2;
    g.M = S; g.N = D; g.K = D;
    g.page_table = reinterpret_cast<float* const*>(page_table); g.lengths = lengths; g.new_batch_idx = new_idx;
    g.B = B; g.S = S;
    g.n_new = n_new; g.compact = fill_compact(n_new);
    dim3 grid(ceil_div_i(D, HN) * 2, ceil_div_i(S, HM), n_new);
    if (g.compact) grid = dim3(ceil_div_i(D, HN) * 2, ceil_div_i(S * n_new, HM), 1);
    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedFill>), grid, dim3(kHThreads), 0, st, g);
    return launch_status();
}

// ---- fp8 (OCP e4m3) pages, bf16 weights: MLI_ELEM_FP8 of the lean entry points --------------------------------------
int launch_latest_paged_fp8(uint8_t* const* page_table, const int* lengths, const uint16_t* wk, const uint16_t* wq,
                            const uint16_t* wv, float* q, int B, int S, int D, hipStream_t st) {
    if (B <= 0 || S % kPage != 0 || D % 16 != 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.w[0] = reinterpret_cast<const float*>(wk); g.w[1] = reinterpret_cast<const float*>(wq);
    g.w[2] = reinterpret_cast<const float*>(wv); g.n_out = 3;
    g.out_id[0] = 0; g.out_id[1] = 1; g.out_id[2] = 2;
    g.M = B; g.N = D; g.K = D;
    g.page_table = reinterpret_cast<float* const*>(page_table); g.q_output = q; g.lengths = lengths;
    g.B = B; g.S = S;
    g.compact = latest_compact(B, D);
    const int tiles_x = ceil_div_i(D, HN) * 3;
    if (gemm_use_tall_tiles((int64_t)tiles_x * ceil_div_i(B, 128))) {
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedLatest, 2, 64, true>), dim3(tiles_x, ceil_div_i(B, 128), 1),
                           dim3(kHThreads), 0, st, g);
        return launch_status();
    }
    dim3 grid(tiles_x, ceil_div_i(B, HM), 1);
    if (D >= 256 && D <= 1024 && deep_k_tiles_enabled())
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedLatest, 1, 128, true>), grid, dim3(kHThreads), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedLatest, 1, 32, true>), grid, dim3(kHThreads), 0, st, g);
    return launch_status();
}

// emb_table == nullptr: the x rows are read from segment 0 (fill only); else the encoder is the GEMM's prologue
int launch_fill_paged_fp8_embed(const float* emb_table, const float* wpe, const int* tokens, uint8_t* const* page_table,
                                const int* new_idx, const int* lengths, const uint16_t* wk, const uint16_t* wv, int B, int S,
                                int D, int n_new, hipStream_t st) {
    if (n_new == 0) return 0;
    if (n_new < 0 || B <= 0 || S % kPage != 0 || D % 16 != 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.emb_table = emb_table; g.wpe = wpe; g.inp = tokens;
    g.w[0] = reinterpret_cast<const float*>(wk); g.w[1] = reinterpret_cast<const float*>(wv); g.n_out = 2;
    g.out_id[0] = 0; g.out_id[1] = 2;
    g.M = S; g.N = D; g.K = D;
    g.page_table = reinterpret_cast<float* const*>(page_table); g.lengths = lengths; g.new_batch_idx = new_idx;
    g.B = B; g.S = S;
    g.n_new = n_new; g.compact = fill_compact(n_new);
    dim3 grid(ceil_div_i(D, HN) * 2, ceil_div_i(S, HM), n_new);
    if (g.compact) grid = dim3(ceil_div_i(D, HN) * 2, ceil_div_i(S * n_new, HM), 1);
    hipLaunchKernelGGL((gemm_bf16_mfma_kernel<kPagedFill, 1, 32, true>), grid, dim3(kHThreads), 0, st, g);
    return launch_status();
}

int launch_fill_paged_bf16_native(uint16_t* const* page_table, const int* new_idx, const int* lengths,
                                  const uint16_t* wk, const uint16_t* wv, int B, int S, int D, int n_new,
                                  hipStream_t st) {
    return launch_fill_paged_bf16_embed(nullptr, nullptr, nullptr, page_table, new_idx, lengths, wk, wv, B, S, D, n_new, st);
}

}  // namespace mli

#ifdef MLI_DMA_TRACE
extern "C" int mli_debug_dma_trace(unsigned long long* host, int n_wgs) {
    if (n_wgs > mli::kDmaTraceWgs) n_wgs = mli::kDmaTraceWgs;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mli::mli_dma_trace), (size_t)n_wgs * 16 * sizeof(unsigned long long));
}
#endif

#ifdef MLI_GEMM_TRACE
extern "C" int mli_debug_gemm_trace(unsigned long long* host, int n_wgs) {
    if (n_wgs > mli::kGemmTraceWgs) n_wgs = mli::kGemmTraceWgs;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mli::mli_gemm_trace), (size_t)n_wgs * 8 * sizeof(unsigned long long));
}
#endif
