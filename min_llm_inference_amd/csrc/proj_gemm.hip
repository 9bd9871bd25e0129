// fp32 MFMA GEMM core for every dense product on the decode path:
//   * decode projection   x[B,D] . [Wk|Wq|Wv]   with gather prologue + page/cache scatter epilogue
//       replaces get_latest_kt_q_v                  (src/kernels/self_attention_inference_optimized.cu:100-143)
//                get_latest_k_q_v_paged_attention   (src/kernels/paged_attention.cu:126-180)
//                get_latest_batch_embs + 3 x cublasSgemm + save_to_page_table
//                                                   (src/kernels/paged_attention_cublas.cu:16-99)
//   * prefill fill        X_b[L_b,D] . [Wk|Wv]  for newly inserted rows
//       replaces fill_new_kt_v_cache                (…optimized.cu:27-85)
//                fill_new_k_v_cache_paged_attention (paged_attention.cu:20-87)
//                fill_new_k_v_cache_paged_attention_warp_tiling (paged_attention_cublas.cu:112-223)
//   * decoder logits      attn[B,D] . emb_table[V,D]^T
//       replaces gemm_transpose_kernel (src/kernels/gemm.cu:13-51) / cublasSgemm (src/kernels/decoder.cu:247-249)
//
// v_mfma_f32_32x32x2_f32 is an exact, k-ordered fp32 fma chain, so results are fp32-faithful
// (no TF32-style truncation).  Tile: 64x64x32 per 256-thread workgroup, 2x2 waves of 32x32.
// Per-row source / destination pointers are resolved once per workgroup into LDS (this is where
// the page table is read: one pointer per row, never inside the k loop).
#include "gemm_common.hpp"

namespace mli {

constexpr int BM = 64, BN = 64;   // (rows per workgroup and the k extent of a staged tile are set inside the kernel)
constexpr int LDB = BN + 4;   // [k][n] tile written with b128 stores: rows stay 16-byte aligned
constexpr int LDBT = BN + 1;  // [n][k] source (transposed B): same treatment as A
constexpr int kGemmThreads = 256;

using f32x16 = __attribute__((ext_vector_type(16))) float;

// VEC4: K % 4 == 0, N % 4 == 0 and 16-byte aligned rows -> float4 global loads.
// BF16 (paged modes only): x, W and the K/V outputs are bfloat16, q and the accumulation fp32.  Operands are
// widened to fp32 while they are staged into LDS, so the product is still the exact fp32 MFMA chain
// (bf16 x bf16 products are exact in fp32); a native v_mfma_f32_32x32x16_bf16 tile is the next step.
// MT: 32-row sub-tiles per wave along M.  MT = 2 (128x64 workgroup tile) shares every weight fragment between two
// MFMAs with independent accumulators: 1.5 instead of 2 LDS fragment reads per MFMA and half the barriers per flop.
// KQ: 32-deep k slabs staged per tile.  Only KQ = 1 is instantiated: 128 k per barrier pair (KQ = 4) was measured
// SLOWER for the latency-bound fp32 shapes (config 4 logits 23.8 -> 26.0 us, D = 2048 logits 61.8 -> 70.2 us) --
// unlike the bf16 kernel, whose deep tile is a win (proj_gemm_bf16.hip).
template <int MODE, bool BT, bool VEC4, bool BF16 = false, int MT = 1, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 2 * kGemmThreads : kGemmThreads) void gemm_f32_mfma_kernel(GemmArgs g) {
    constexpr int BM = 64 * MT;   // shadows the namespace-level 64: rows per workgroup
    constexpr int BK = 32;        // k extent of a staged tile
    constexpr int KQ = 1;
    constexpr int LDA = BM + 1;
    constexpr int kBufs = SPLIT ? 2 : 1;
    constexpr int kThreads = SPLIT ? 2 * kGemmThreads : kGemmThreads;
    __shared__ float As[kBufs * BK * LDA];
    constexpr int LDBX = BT ? LDBT : LDB;
    __shared__ __align__(16) float Bs[kBufs * BK * LDBX];
    __shared__ const float* a_ptr[BM];
    __shared__ float* o_ptr[BM];
    constexpr bool kFillMode = MODE == kNaiveFill || MODE == kPagedFill;
    __shared__ const float* e_ptr[kFillMode ? BM : 1];  // embedding prologue: emb_table / wpe row of every A row
    __shared__ const float* p_ptr[kFillMode ? BM : 1];

    const int tiles_n = (g.N + BN - 1) / BN;
    const int wsel = blockIdx.x / tiles_n;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int m0 = blockIdx.y * BM;
    const int z = blockIdx.z;
    const int out_id = g.out_id[wsel];
    const float* __restrict__ Bmat = g.w[wsel];

    constexpr bool kFill = MODE == kNaiveFill || MODE == kPagedFill;
    constexpr bool kLatest = MODE == kNaiveLatest || MODE == kPagedLatest;
    __shared__ std::conditional_t<kFill || kLatest, FillIndex, NoFillIndex> fill_index[1];
    const int tid = threadIdx.x;
    int fill_total = 0;
    if (kFill || kLatest) {
        if (g.compact) {
            fill_total = build_fill_index<kThreads, kLatest>(g, fill_index[0]);
            if (m0 >= fill_total) return;  // workgroup-uniform
        } else if (kFill && m0 >= g.lengths[g.new_batch_idx[z]]) {
            return;  // whole tile beyond the row's length: nothing to do (reference …optimized.cu:43-45)
        }
    }

    const bool embed = kFillMode && !BF16 && g.emb_table != nullptr;
    if (tid < BM) {  // BM <= 128 < 256 threads
        RowDesc r{nullptr, nullptr, nullptr, nullptr};
        if ((kFill || kLatest) && g.compact) {
            if (m0 + tid < fill_total) {
                int zz, ss;
                fill_index_lookup(fill_index[0], kLatest ? g.B : g.n_new, m0 + tid, zz, ss);
                r = kLatest ? resolve_row<MODE, BF16>(g, zz, 0, out_id) : resolve_row<MODE, BF16>(g, ss, zz, out_id);
            }
        } else {
            r = resolve_row<MODE, BF16>(g, m0 + tid, z, out_id);
        }
        a_ptr[tid] = r.a;
        o_ptr[tid] = r.o;
        if (kFillMode) {
            e_ptr[tid] = r.e;
            p_ptr[tid] = r.p;
        }
    }
    // only the contiguous layout keeps K transposed (kt_cache[b, n, s]): element stride S along n
    constexpr bool kCanTranspose = MODE == kNaiveLatest || MODE == kNaiveFill;
    const bool transposed_out = kCanTranspose && out_id == 0;
    const int64_t o_stride = transposed_out ? g.S : 1;
    __syncthreads();

    // SPLIT: threads 0..255 (waves 0-3) multiply, threads 256..511 (waves 4-7) move the next tile into the other LDS buffer
    const bool is_loader = SPLIT && tid >= kGemmThreads;
    const int st = is_loader ? tid - kGemmThreads : tid;   // index among the staging threads
    const int lane = tid & 63;
    const int wave = (tid >> 6) & 3;
    const int wm = (wave >> 1) * 32 * MT;
    const int wn = (wave & 1) * 32;

    // global -> register staging coordinates
    //   A tile [BM][BK]: 8 threads per row (float4 along k), 32 rows per pass, 2 passes
    //   B tile [BK][BN]: 16 threads per k-row (float4 along n), 16 k-rows per pass, 2 passes
    //   B^T tile (BT): rows are n, float4 along k -- same shape as the A tile
    const int a_row = st >> 3, a_kq = (st & 7) * 4;
    const int b_row = st >> 4, b_nq = (st & 15) * 4;
    constexpr int AP = 2 * MT;  // A staging passes of 32 rows
    float4 a_regs[KQ][AP], b_regs[KQ][2];
    // this thread's source rows, kept in registers: re-reading them from LDS every tile put a wait-for-LDS and a
    // branch per pass into every k step
    const float* a_src[AP];
    const float* e_src[AP];
    const float* p_src[AP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        a_src[p] = a_ptr[a_row + p * 32];
        e_src[p] = kFillMode ? e_ptr[a_row + p * 32] : nullptr;
        p_src[p] = kFillMode ? p_ptr[a_row + p * 32] : nullptr;
    }
    const bool writes_x = blockIdx.x == 0;  // first column tile of the first weight: every A element passes once

    auto load_slab = [&](int k0, float4 (&a_reg)[AP], float4 (&b_reg)[2]) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const float* ap = a_src[p];
            const int k = k0 + a_kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kFillMode && embed && ap != nullptr) {
                // encoder as prologue: x = emb[tok] + wpe[s], the sum the encoder kernel writes (one fp32 add)
                if (VEC4) {
                    if (k < g.K) {
                        const float4 e4 = *reinterpret_cast<const float4*>(e_src[p] + k);
                        const float4 p4 = *reinterpret_cast<const float4*>(p_src[p] + k);
                        v = make_float4(e4.x + p4.x, e4.y + p4.y, e4.z + p4.z, e4.w + p4.w);
                        if (writes_x) *reinterpret_cast<float4*>(const_cast<float*>(ap) + k) = v;
                    }
                } else {
                    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (k + i < g.K) {
                            t[i] = e_src[p][k + i] + p_src[p][k + i];
                            if (writes_x) const_cast<float*>(ap)[k + i] = t[i];
                        }
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
            } else if (ap != nullptr) {
                if (BF16) {
                    if (k < g.K) v = load4_bf16(reinterpret_cast<const uint16_t*>(ap) + k);
                } else if (VEC4) {
                    if (k < g.K) v = *reinterpret_cast<const float4*>(ap + k);
                } else {
                    if (k + 0 < g.K) v.x = ap[k + 0];
                    if (k + 1 < g.K) v.y = ap[k + 1];
                    if (k + 2 < g.K) v.z = ap[k + 2];
                    if (k + 3 < g.K) v.w = ap[k + 3];
                }
            }
            a_reg[p] = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BT) {
                const int n = n0 + a_row + p * 32;
                const int k = k0 + a_kq;
                if (n < g.N) {
                    const float* bp = Bmat + (int64_t)n * g.K + k;
                    if (VEC4) {
                        if (k < g.K) v = *reinterpret_cast<const float4*>(bp);
                    } else {
                        if (k + 0 < g.K) v.x = bp[0];
                        if (k + 1 < g.K) v.y = bp[1];
                        if (k + 2 < g.K) v.z = bp[2];
                        if (k + 3 < g.K) v.w = bp[3];
                    }
                }
            } else {
                const int k = k0 + b_row + p * 16;
                const int n = n0 + b_nq;
                if (k < g.K) {
                    const float* bp = Bmat + (int64_t)k * g.N + n;
                    if (BF16) {
                        if (n < g.N) v = load4_bf16(reinterpret_cast<const uint16_t*>(Bmat) + (int64_t)k * g.N + n);
                    } else if (VEC4) {
                        if (n < g.N) v = *reinterpret_cast<const float4*>(bp);
                    } else {
                        if (n + 0 < g.N) v.x = bp[0];
                        if (n + 1 < g.N) v.y = bp[1];
                        if (n + 2 < g.N) v.z = bp[2];
                        if (n + 3 < g.N) v.w = bp[3];
                    }
                }
            }
            b_reg[p] = v;
        }
    };
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) load_slab(k0 + q * 32, a_regs[q], b_regs[q]);
    };

    auto store_slab = [&](float* As_q, float* Bs_q, const float4 (&a_reg)[AP], const float4 (&b_reg)[2]) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int m = a_row + p * 32;
            As_q[(a_kq + 0) * LDA + m] = a_reg[p].x;
            As_q[(a_kq + 1) * LDA + m] = a_reg[p].y;
            As_q[(a_kq + 2) * LDA + m] = a_reg[p].z;
            As_q[(a_kq + 3) * LDA + m] = a_reg[p].w;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (BT) {
                const int n = a_row + p * 32;
                Bs_q[(a_kq + 0) * LDBX + n] = b_reg[p].x;
                Bs_q[(a_kq + 1) * LDBX + n] = b_reg[p].y;
                Bs_q[(a_kq + 2) * LDBX + n] = b_reg[p].z;
                Bs_q[(a_kq + 3) * LDBX + n] = b_reg[p].w;
            } else {
                *reinterpret_cast<float4*>(&Bs_q[(b_row + p * 16) * LDBX + b_nq]) = b_reg[p];
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int q = 0; q < KQ; ++q) store_slab(As + q * 32 * LDA, Bs + q * 32 * LDBX, a_regs[q], b_regs[q]);
    };

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;

    const int nk = (g.K + BK - 1) / BK;
    const int lk = lane >> 5;   // which of the 2 k's of an MFMA step this lane feeds
    const int li = lane & 31;   // row (A) / column (B) inside the 32x32 wave tile

    auto multiply = [&](const float* A, const float* B) {
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float b = B[(kk + lk) * LDBX + wn + li];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float a = A[(kk + lk) * LDA + wm + mt * 32 + li];
                // the K output of the contiguous layout is stored transposed: swap operands so
                // that the sequence index lands on the lane (coalesced kt_cache stores)
                if (transposed_out) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[mt], 0, 0, 0);
                else acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt], 0, 0, 0);
            }
        }
    };
    if constexpr (!SPLIT) {
        load_tile(0);
        store_tile();
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) load_tile((t + 1) * BK);  // in flight under the MFMAs below
            multiply(As, Bs);
            __syncthreads();
            if (t + 1 < nk) {
                store_tile();
                __syncthreads();
            }
        }
    } else {
        // One wave doing everything in turn (issue loads, wait, write LDS, read fragments, multiply) is overlapped only by
        // the other waves of its SIMD; with about one workgroup per CU (logits of 1024 rows, prefill of a few hundred
        // tokens: 256 tiles of 64 x 64) nothing overlaps and the matrix pipe idles half the time.  Here four waves only load
        // (two tiles in flight in their registers) and four only multiply, one barrier per k step -- the structure round 2
        // found on the bf16 projection (whose loaders have since become LDS-DMA issuers, proj_gemm_bf16.hip).  Same MFMA
        // chain per output element: bit-identical results.
        float* As1 = As + BK * LDA;
        float* Bs1 = Bs + BK * LDBX;
        if (is_loader) {
            float4 a2[AP], b2[2];
            load_slab(0, a_regs[0], b_regs[0]);
            if (1 < nk) load_slab(BK, a2, b2);
            store_slab(As, Bs, a_regs[0], b_regs[0]);
            if (2 < nk) load_slab(2 * BK, a_regs[0], b_regs[0]);
            __syncthreads();  // tile 0 is in buffer 0
            for (int t = 0; t < nk; t += 2) {
                if (t + 1 < nk) store_slab(As1, Bs1, a2, b2);                       // tile t + 1
                if (t + 3 < nk) load_slab((t + 3) * BK, a2, b2);
                __syncthreads();
                if (t + 1 < nk) {
                    if (t + 2 < nk) store_slab(As, Bs, a_regs[0], b_regs[0]);      // tile t + 2
                    if (t + 4 < nk) load_slab((t + 4) * BK, a_regs[0], b_regs[0]);
                    __syncthreads();
                }
            }
        } else {
            __syncthreads();  // tile 0 is in buffer 0
            for (int t = 0; t < nk; t += 2) {
                multiply(As, Bs);
                __syncthreads();
                if (t + 1 < nk) {
                    multiply(As1, Bs1);
                    __syncthreads();
                }
            }
        }
    }

    // epilogue: accumulator register r of lane l is tile element
    //   (row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31)
    if (MODE == kPlain && g.row_best != nullptr) {
        // argmax epilogue (decoder logits): a row of the wave's 32x32 sub-tile lies across 32 lanes of one register
        // -> butterfly over the 5 low lane bits; the two waves that share the rows meet in LDS (the staging tiles are
        // free after the k loop's last barrier); one (max, index) pair per (row, column tile) goes to memory
        float* best_v = As;                                   // [BM][2]
        int* best_i = reinterpret_cast<int*>(As + BM * 2);    // [BM][2]   (BK * LDA >= 4 * BM floats)
        const int n = n0 + wn + li;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (is_loader) break;  // (wave-uniform: the loader waves only meet the barrier below)
            // a score takes part only if it beats -FLT_MAX, as in decoder_argmax_kernel (NaN and -inf never win)
            float v[16];
            int ix[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool takes_part = n < g.N && acc[mt][r] > -3.402823466e+38f;
                v[r] = takes_part ? acc[mt][r] : -3.402823466e+38f;
                ix[r] = takes_part ? n : -1;
            }
            // Transposing butterfly over the 32 lanes that hold one row per register: at every step a lane keeps half
            // of its rows and trades the other half with its partner (15 exchanges instead of 16 x 5), branch-free,
            // so the exchanges of different rows overlap.  Afterwards lane li holds row register (li >> 1) & 15.
            argmax_butterfly_step<8>(v, ix, (lane & 16) != 0, 16);
            argmax_butterfly_step<4>(v, ix, (lane & 8) != 0, 8);
            argmax_butterfly_step<2>(v, ix, (lane & 4) != 0, 4);
            argmax_butterfly_step<1>(v, ix, (lane & 2) != 0, 2);
            {
                const float ov = __shfl_xor(v[0], 1, kWave);
                const int oi = __shfl_xor(ix[0], 1, kWave);
                argmax_take(v[0], ix[0], ov, oi);
            }
            if ((li & 1) == 0) {
                const int r = (li >> 1) & 15;
                const int row = wm + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                best_v[row * 2 + (wave & 1)] = v[0];
                best_i[row * 2 + (wave & 1)] = ix[0];
            }
        }
        __syncthreads();
        if (!is_loader && tid < BM && m0 + tid < g.M) {
            float v = best_v[tid * 2];
            int i = best_i[tid * 2];
            argmax_take(v, i, best_v[tid * 2 + 1], best_i[tid * 2 + 1]);
            g.row_best[(int64_t)(m0 + tid) * tiles_n + blockIdx.x % tiles_n] = RowBest{v, i};
        }
        return;
    }
    if (is_loader) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int trow = (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (transposed_out) {
                const int n = n0 + wn + trow;   // rows of the swapped product run along N
                const int mi = wm + mt * 32 + li;
                float* op = o_ptr[mi];
                if (op != nullptr && n < g.N) op[(int64_t)n * o_stride] = acc[mt][r];
            } else {
                const int mi = wm + mt * 32 + trow;
                const int n = n0 + wn + li;
                float* op = o_ptr[mi];
                if (op != nullptr && n < g.N) {
                    if (BF16 && out_id != 1) reinterpret_cast<uint16_t*>(op)[n] = f32_to_bf16(acc[mt][r]);
                    else op[n] = acc[mt][r];
                }
            }
        }
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static thread_local int g_fill_compact = 1;  // mli_tune "fill_compact": 0 = one tile grid per new row (the reference's decomposition)
void set_fill_compact(int v) { g_fill_compact = v != 0; }
int fill_compact(int n_new) { return g_fill_compact && n_new <= kMaxCompactRows ? 1 : 0; }
// mli_tune "latest_compact": 0 = the decode projection multiplies empty rows as zeros, 1 (default) = it multiplies a
// device-built list of the non-empty rows where that can pay, 2 = wherever possible (tests).  Every workgroup rebuilds the
// list (a prefix sum over the batch rows): ~1.8 us of prologue -- config 4's projection (emb_dim 512) takes 9.4 us without
// it and 11.2 with it, and could save 3-4 us at best from a batch that is 40 % empty slots; at emb_dim 2048 (222 us) the same
// batch saves 90 us.  So: only for reductions of 1024 and more.
static thread_local int g_latest_compact = 1;
void set_latest_compact(int v) { g_latest_compact = v < 0 ? 0 : (v > 2 ? 2 : v); }
int latest_compact(int n_batch, int k_dim) {
    if (g_latest_compact == 0 || n_batch > kMaxCompactRows) return 0;
    return g_latest_compact == 2 || k_dim >= 1024 ? 1 : 0;
}

static thread_local int g_deep_k_tiles = 1;  // mli_tune "gemm_deep_k" (bf16 kernel): 0 = 32-deep staged tiles everywhere
void set_deep_k_tiles(int v) { g_deep_k_tiles = v != 0; }
int deep_k_tiles_enabled() { return g_deep_k_tiles; }
// mli_tune "prefill_fused": which form mli_[paged_]prefill runs: 1 (default) = the encoder as the fill GEMM's prologue up to
// emb_dim 512 and encoder + fill as two launches beyond, 0 = always the two launches, 2 = always the prologue form
static thread_local int g_prefill_fused = 1;
void set_prefill_fused(int v) { g_prefill_fused = v < 0 ? 0 : (v > 2 ? 2 : v); }
bool prefill_fuses(int emb_dim) { return g_prefill_fused == 2 || (g_prefill_fused == 1 && emb_dim <= 512); }

static thread_local int g_gemm_split = 1;  // mli_tune "gemm_split": 0 = never the loader / MFMA wave split of the 64-row-tile kernel
void set_gemm_split(int v) { g_gemm_split = v != 0; }
static thread_local int g_gemm_tall_tiles = 1;  // mli_tune "gemm_tall_tiles": 0 = always 64-row tiles, 2 = 128-row tiles whenever allowed (tests)
void set_gemm_tall_tiles(int v) { g_gemm_tall_tiles = v < 0 ? 0 : (v > 2 ? 2 : v); }
bool gemm_use_tall_tiles(int64_t tall_workgroups) { return g_gemm_tall_tiles == 2 || (g_gemm_tall_tiles == 1 && tall_workgroups >= 512); }

// proj_gemm_panel.hip: the latency-shaped kernel for small decode-step products
bool gemm_panel_wanted(int M, int N_total, int K, bool vec4);
int gemm_panel_tiles_n(int N);
template <int MODE, bool BT>
int launch_gemm_panel(const GemmArgs& g, int rows, hipStream_t st);

// rows = live extent of the M dimension (per z-slice)
template <int MODE, bool BT>
static int launch_gemm(const GemmArgs& g, int rows, int z, bool vec4, hipStream_t st) {
    if (g.N <= 0 || g.K <= 0 || rows <= 0 || z <= 0) return MLI_ERR_BAD_ARG;
    constexpr bool kPanelMode = ((MODE == kPagedLatest || MODE == kNaiveLatest) && !BT) || (MODE == kPlain && BT);
    if constexpr (kPanelMode) {
        if (z == 1 && gemm_panel_wanted(rows, g.N * g.n_out, g.K, vec4)) return launch_gemm_panel<MODE, BT>(g, rows, st);
    }
    const int tiles_x = ceil_div_i(g.N, BN) * g.n_out;
    // 128-row tiles when they still fill the chip (>= 2 workgroups per CU) -- decode projection / logits of a large
    // batch; the prefill fill keeps 64-row tiles (a new row's prompt rarely fills 128 rows)
    constexpr bool kTallOk = MODE == kPagedLatest || MODE == kNaiveLatest || MODE == kPlain;
    if (kTallOk && vec4 && gemm_use_tall_tiles((int64_t)tiles_x * ceil_div_i(rows, 128) * z)) {
        dim3 grid(tiles_x, ceil_div_i(rows, 128), z);
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE, BT, true, false, 2>), grid, dim3(kGemmThreads), 0, st, g);
        return launch_status();
    }
    dim3 grid(tiles_x, ceil_div_i(rows, BM), z);
    if (g.compact && (MODE == kNaiveFill || MODE == kPagedFill))
        grid = dim3(tiles_x, ceil_div_i(rows * z, BM), 1);  // flat (new row, token) list: upper bound
    // a reduction long enough to pipeline: loader waves + MFMA waves (512 threads).  Measured at D = 2048: logits of 1024
    // rows 61 -> 52 us, prefill of 128 / 256 / 512 / 4096 prompt tokens 60 -> 49 / 63 -> 52 / 96 -> 86 / 593 -> 588 us,
    // config 4 logits (K = 512) 21.4 -> 19.5 us: never slower
    if (vec4 && g_gemm_split && g.K >= 256) {
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE, BT, true, false, 1, true>), grid, dim3(2 * kGemmThreads), 0, st, g);
        return launch_status();
    }
    if (vec4) hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE, BT, true>), grid, dim3(kGemmThreads), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE, BT, false>), grid, dim3(kGemmThreads), 0, st, g);
    return launch_status();
}

int launch_latest_naive(const float* inp, const int* lengths, const float* wk, const float* wq, const float* wv,
                        float* kt, float* v, float* q, int B, int S, int Din, int Dout, hipStream_t st) {
    if (B <= 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.w[0] = wk; g.w[1] = wq; g.w[2] = wv; g.n_out = 3;
    g.out_id[0] = 0; g.out_id[1] = 1; g.out_id[2] = 2;
    g.M = B; g.N = Dout; g.K = Din;
    g.inp_embedding = inp; g.kt_cache = kt; g.v_cache = v; g.q_output = q; g.lengths = lengths;
    g.B = B; g.S = S;
    g.compact = latest_compact(B, Din);
    const bool vec4 = Din % 4 == 0 && Dout % 4 == 0 && aligned16(inp) && aligned16(wk) && aligned16(wq) && aligned16(wv);
    return launch_gemm<kNaiveLatest, false>(g, B, 1, vec4, st);
}

// emb_table != nullptr: the encoder runs as the GEMM's prologue (inp_embedding rows are written, not read)
int launch_fill_naive_embed(const float* emb_table, const float* wpe, const int* tokens, float* inp, const int* new_idx,
                            const int* lengths, const float* wk, const float* wv, float* kt, float* v, int B, int S,
                            int Din, int Dout, int n_new, hipStream_t st) {
    if (n_new == 0) return 0;  // reference …optimized.cu:308-310
    if (n_new < 0 || B <= 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.emb_table = emb_table; g.wpe = wpe; g.inp = tokens;
    g.w[0] = wk; g.w[1] = wv; g.n_out = 2;
    g.out_id[0] = 0; g.out_id[1] = 2;
    g.M = S; g.N = Dout; g.K = Din;
    g.inp_embedding = inp; g.kt_cache = kt; g.v_cache = v; g.lengths = lengths; g.new_batch_idx = new_idx;
    g.B = B; g.S = S;
    g.n_new = n_new; g.compact = fill_compact(n_new);
    const bool vec4 = Din % 4 == 0 && Dout % 4 == 0 && aligned16(inp) && aligned16(wk) && aligned16(wv) &&
                      (emb_table == nullptr || (aligned16(emb_table) && aligned16(wpe)));
    return launch_gemm<kNaiveFill, false>(g, S, n_new, vec4, st);
}

int launch_fill_naive(const float* inp, const int* new_idx, const int* lengths, const float* wk, const float* wv,
                      float* kt, float* v, int B, int S, int Din, int Dout, int n_new, hipStream_t st) {
    return launch_fill_naive_embed(nullptr, nullptr, nullptr, const_cast<float*>(inp), new_idx, lengths, wk, wv, kt, v, B,
                                   S, Din, Dout, n_new, st);
}

int launch_latest_paged(float* const* page_table, const int* lengths, const float* wk, const float* wq,
                        const float* wv, float* q, int B, int S, int D, hipStream_t st) {
    if (B <= 0 || S % kPage != 0 || D % 4 != 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.w[0] = wk; g.w[1] = wq; g.w[2] = wv; g.n_out = 3;
    g.out_id[0] = 0; g.out_id[1] = 1; g.out_id[2] = 2;
    g.M = B; g.N = D; g.K = D;
    g.page_table = page_table; g.q_output = q; g.lengths = lengths;
    g.B = B; g.S = S;
    g.compact = latest_compact(B, D);
    const bool vec4 = aligned16(wk) && aligned16(wq) && aligned16(wv);
    return launch_gemm<kPagedLatest, false>(g, B, 1, vec4, st);
}

int launch_fill_paged_embed(const float* emb_table, const float* wpe, const int* tokens, float* const* page_table,
                            const int* new_idx, const int* lengths, const float* wk, const float* wv, int B, int S, int D,
                            int n_new, hipStream_t st) {
    if (n_new == 0) return 0;  // reference paged_attention.cu:100-102
    if (n_new < 0 || B <= 0 || S % kPage != 0 || D % 4 != 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.emb_table = emb_table; g.wpe = wpe; g.inp = tokens;
    g.w[0] = wk; g.w[1] = wv; g.n_out = 2;
    g.out_id[0] = 0; g.out_id[1] = 2;
    g.M = S; g.N = D; g.K = D;
    g.page_table = page_table; g.lengths = lengths; g.new_batch_idx = new_idx;
    g.B = B; g.S = S;
    g.n_new = n_new; g.compact = fill_compact(n_new);
    const bool vec4 = aligned16(wk) && aligned16(wv) && (emb_table == nullptr || (aligned16(emb_table) && aligned16(wpe)));
    return launch_gemm<kPagedFill, false>(g, S, n_new, vec4, st);
}

int launch_fill_paged(float* const* page_table, const int* new_idx, const int* lengths, const float* wk,
                      const float* wv, int B, int S, int D, int n_new, hipStream_t st) {
    return launch_fill_paged_embed(nullptr, nullptr, nullptr, page_table, new_idx, lengths, wk, wv, B, S, D, n_new, st);
}

// bf16 tile engine: 1 = native v_mfma_f32_32x32x16_bf16 (proj_gemm_bf16.hip, default), 0 = operands widened to
// fp32 in LDS + fp32 MFMA (bit-identical to a sequential fp32 sum; kept for parity checks).  mli_tune knob.
static thread_local int g_bf16_native_mfma = 1;
void set_bf16_native_mfma(int v) { g_bf16_native_mfma = v != 0; }
int launch_latest_paged_bf16_native(uint16_t* const*, const int*, const uint16_t*, const uint16_t*, const uint16_t*,
                                    float*, int, int, int, hipStream_t);
int launch_fill_paged_bf16_native(uint16_t* const*, const int*, const int*, const uint16_t*, const uint16_t*, int, int,
                                  int, int, hipStream_t);

int launch_latest_paged_bf16(uint16_t* const* page_table, const int* lengths, const uint16_t* wk,
                             const uint16_t* wq, const uint16_t* wv, float* q, int B, int S, int D, hipStream_t st) {
    if (B <= 0 || S % kPage != 0 || D % 8 != 0) return MLI_ERR_BAD_ARG;
    if (g_bf16_native_mfma) return launch_latest_paged_bf16_native(page_table, lengths, wk, wq, wv, q, B, S, D, st);
    GemmArgs g{};
    g.w[0] = reinterpret_cast<const float*>(wk); g.w[1] = reinterpret_cast<const float*>(wq);
    g.w[2] = reinterpret_cast<const float*>(wv); g.n_out = 3;
    g.out_id[0] = 0; g.out_id[1] = 1; g.out_id[2] = 2;
    g.M = B; g.N = D; g.K = D;
    g.page_table = reinterpret_cast<float* const*>(page_table); g.q_output = q; g.lengths = lengths;
    g.B = B; g.S = S;
    g.compact = latest_compact(B, D);
    dim3 grid(ceil_div_i(D, BN) * 3, ceil_div_i(B, BM), 1);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<kPagedLatest, false, true, true>), grid, dim3(kGemmThreads), 0, st, g);
    return launch_status();
}

int launch_fill_paged_bf16(uint16_t* const* page_table, const int* new_idx, const int* lengths, const uint16_t* wk,
                           const uint16_t* wv, int B, int S, int D, int n_new, hipStream_t st) {
    if (n_new == 0) return 0;
    if (n_new < 0 || B <= 0 || S % kPage != 0 || D % 8 != 0) return MLI_ERR_BAD_ARG;
    if (g_bf16_native_mfma) return launch_fill_paged_bf16_native(page_table, new_idx, lengths, wk, wv, B, S, D, n_new, st);
    GemmArgs g{};
    g.w[0] = reinterpret_cast<const float*>(wk); g.w[1] = reinterpret_cast<const float*>(wv); g.n_out = 2;
    g.out_id[0] = 0; g.out_id[1] = 2;
    g.M = S; g.N = D; g.K = D;
    g.page_table = reinterpret_cast<float* const*>(page_table); g.lengths = lengths; g.new_batch_idx = new_idx;
    g.B = B; g.S = S;
    g.n_new = n_new; g.compact = fill_compact(n_new);
    dim3 grid(ceil_div_i(D, BN) * 2, ceil_div_i(S, BM), n_new);
    if (g.compact) grid = dim3(ceil_div_i(D, BN) * 2, ceil_div_i(S * n_new, BM), 1);
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<kPagedFill, false, true, true>), grid, dim3(kGemmThreads), 0, st, g);
    return launch_status();
}

// C[M, N] = A[M, K] . Bt[N, K]^T
int launch_gemm_nt(const float* A, const float* Bt, float* C, int M, int N, int K, hipStream_t st) {
    if (M <= 0) return MLI_ERR_BAD_ARG;
    GemmArgs g{};
    g.w[0] = Bt; g.n_out = 1; g.out_id[0] = 1;
    g.M = M; g.N = N; g.K = K;
    g.a_plain = A; g.c_plain = C; g.lda = K; g.ldc = N;
    const bool vec4 = K % 4 == 0 && aligned16(A) && aligned16(Bt);
    return launch_gemm<kPlain, true>(g, M, 1, vec4, st);
}

// The same product with the argmax epilogue: nothing of C[M, N] is stored, row_best[m][t] = (max, lowest index of the
// max) over the columns of tile t.  Tiles per row: gemm_nt_argmax_tiles(N).
// (the panel kernel's tiles are 32 columns wide, the tiled kernel's 64: n_tiles tells the caller which ran)
int gemm_nt_argmax_max_tiles(int N) { return ceil_div_i(N, 32); }
int launch_gemm_nt_argmax(const float* A, const float* Bt, RowBest* row_best, int M, int N, int K, int* n_tiles,
                          hipStream_t st) {
    if (M <= 0 || row_best == nullptr) return MLI_ERR_BAD_ARG;
    {
        const bool vec4 = K % 4 == 0 && aligned16(A) && aligned16(Bt);
        *n_tiles = gemm_panel_wanted(M, N, K, vec4) ? gemm_panel_tiles_n(N) : ceil_div_i(N, BN);
    }
    GemmArgs g{};
    g.w[0] = Bt; g.n_out = 1; g.out_id[0] = 1;
    g.M = M; g.N = N; g.K = K;
    g.a_plain = A; g.c_plain = nullptr; g.lda = K; g.ldc = N;
    g.row_best = row_best;
    const bool vec4 = K % 4 == 0 && aligned16(A) && aligned16(Bt);
    return launch_gemm<kPlain, true>(g, M, 1, vec4, st);
}

}  // namespace mli

extern "C" {

int mli_fill_new_kt_v_cache(const float* inp_embedding, const int* new_batch_idx, const int* lengths,
                            const float* wk, const float* wv, float* kt_cache, float* v_cache, int n_batch,
                            int n_sequence, int input_dim, int output_dim, int n_new_items, void* stream) {
    return mli::launch_fill_naive(inp_embedding, new_batch_idx, lengths, wk, wv, kt_cache, v_cache, n_batch,
                                  n_sequence, input_dim, output_dim, n_new_items, mli::as_stream(stream));
}

int mli_get_latest_kt_q_v(const float* inp_embedding, const int* lengths, const float* wk, const float* wq,
                          const float* wv, float* kt_cache, float* v_cache, float* q_output, int n_batch,
                          int n_sequence, int input_dim, int output_dim, void* stream) {
    return mli::launch_latest_naive(inp_embedding, lengths, wk, wq, wv, kt_cache, v_cache, q_output, n_batch,
                                    n_sequence, input_dim, output_dim, mli::as_stream(stream));
}

int mli_fill_new_k_v_cache_paged(float* const* page_table, const int* new_batch_idx, const int* lengths,
                                 const float* wk, const float* wv, int n_batch, int n_sequence, int emb_dim,
                                 int n_new_items, void* stream) {
    return mli::launch_fill_paged(page_table, new_batch_idx, lengths, wk, wv, n_batch, n_sequence, emb_dim,
                                  n_new_items, mli::as_stream(stream));
}

int mli_get_latest_k_q_v_paged(float* const* page_table, const int* lengths, const float* wk, const float* wq,
                               const float* wv, float* q_output, int n_batch, int n_sequence, int emb_dim,
                               void* stream) {
    return mli::launch_latest_paged(page_table, lengths, wk, wq, wv, q_output, n_batch, n_sequence, emb_dim,
                                    mli::as_stream(stream));
}

int mli_fill_new_k_v_cache_paged_bf16(mli_bf16* const* page_table, const int* new_batch_idx, const int* lengths,
                                      const mli_bf16* wk, const mli_bf16* wv, int n_batch, int n_sequence,
                                      int emb_dim, int n_new_items, void* stream) {
    return mli::launch_fill_paged_bf16(page_table, new_batch_idx, lengths, wk, wv, n_batch, n_sequence, emb_dim,
                                       n_new_items, mli::as_stream(stream));
}

int mli_get_latest_k_q_v_paged_bf16(mli_bf16* const* page_table, const int* lengths, const mli_bf16* wk,
                                    const mli_bf16* wq, const mli_bf16* wv, float* q_output, int n_batch,
                                    int n_sequence, int emb_dim, void* stream) {
    return mli::launch_latest_paged_bf16(page_table, lengths, wk, wq, wv, q_output, n_batch, n_sequence, emb_dim,
                                         mli::as_stream(stream));
}

}  // extern "C"

extern "C" {

int mli_prefill(const float* emb_table, const float* wpe, const int* inp, float* inp_embedding, const int* lengths,
                const int* new_item_indices, const float* wk, const float* wv, float* kt_cache, float* v_cache,
                int n_batch, int n_sequence, int input_dim, int output_dim, int n_new_items, void* stream) {
    if (emb_table == nullptr || wpe == nullptr || inp == nullptr) return MLI_ERR_BAD_ARG;
    if (!mli::prefill_fuses(input_dim)) {   // wide models: encoder + fill as two launches (see mli_paged_prefill)
        int rc = mli_inference_optimized_encoder(emb_table, wpe, inp, inp_embedding, lengths, new_item_indices, n_batch,
                                                 n_sequence, input_dim, n_new_items, stream);
        if (rc) return rc;
        return mli_fill_new_kt_v_cache(inp_embedding, new_item_indices, lengths, wk, wv, kt_cache, v_cache, n_batch,
                                       n_sequence, input_dim, output_dim, n_new_items, stream);
    }
    return mli::launch_fill_naive_embed(emb_table, wpe, inp, inp_embedding, new_item_indices, lengths, wk, wv, kt_cache,
                                        v_cache, n_batch, n_sequence, input_dim, output_dim, n_new_items,
                                        mli::as_stream(stream));
}

}  // extern "C"
