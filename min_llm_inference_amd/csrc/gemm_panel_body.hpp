// The panel GEMM's workgroup body (see proj_gemm_panel.hip for what the kernel is and why).
#pragma once

#include "gemm_common.hpp"

namespace mli {

constexpr int PM = 32, PN = 32;  // workgroup tile
constexpr int KP = 256;          // k extent of one staged panel
constexpr int LDK = KP + 2;      // floats per LDS row: (2 * row + k) mod 64 distinct over a 16-row x 4-k fragment
constexpr int kPanelThreads = 256;

using f32x4 = __attribute__((ext_vector_type(4))) float;

// B operand source: BT = false: w[k][n] (projection weights), BT = true: w[n][k] (embedding table, logits)
// bx, by: the tile's grid position (gemm_f32_panel_kernel: blockIdx.x, blockIdx.y).
template <int MODE, bool BT>
__device__ __forceinline__ void gemm_panel_tile(const GemmArgs& g, int bx, int by, unsigned char* panel_smem) {
    float* As = reinterpret_cast<float*>(panel_smem);   // [PM][LDK]
    float* Bs = As + PM * LDK;                           // [PN][LDK]
    const float** a_ptr = reinterpret_cast<const float**>(Bs + PN * LDK);  // [PM]
    float** o_ptr = reinterpret_cast<float**>(Bs + PN * LDK) + PM;         // [PM]

    const int tiles_n = (g.N + PN - 1) / PN;
    const int wsel = bx / tiles_n;
    const int tile = bx % tiles_n;
    const int n0 = tile * PN;
    const int m0 = by * PM;
    const int out_id = g.out_id[wsel];
    const float* __restrict__ Bmat = g.w[wsel];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // ---- B panel requests first: they depend on nothing but the grid position ----
    // A-shaped staging (also B when BT): 8 passes of 4 rows, a wave reads one row's 1 KiB per pass
    const int r_row = tid >> 6, r_kq = (tid & 63) * 4;
    // [k][n] staging (B, not BT): 8 passes of 32 k rows, 8 threads per 128-byte row
    const int b_k = tid >> 3, b_nq = (tid & 7) * 4;
    float4 a_regs[8], b_regs[8];
    auto load_b = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BT) {
                const int n = n0 + p * 4 + r_row, k = k0 + r_kq;
                if (n < g.N && k < g.K) v = *reinterpret_cast<const float4*>(Bmat + (int64_t)n * g.K + k);
            } else {
                const int k = k0 + p * 32 + b_k, n = n0 + b_nq;
                if (k < g.K && n < g.N) v = *reinterpret_cast<const float4*>(Bmat + (int64_t)k * g.N + n);
            }
            b_regs[p] = v;
        }
    };
    load_b(0);

    // ---- rows of this tile: source and destination pointers, once per workgroup ----
    if (tid < PM) {
        const RowDesc r = resolve_row<MODE, false>(g, m0 + tid, 0, out_id);
        a_ptr[tid] = r.a;
        o_ptr[tid] = r.o;
    }
    __syncthreads();
    const float* a_src[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) a_src[p] = a_ptr[p * 4 + r_row];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int k = k0 + r_kq;
            if (a_src[p] != nullptr && k < g.K) v = *reinterpret_cast<const float4*>(a_src[p] + k);
            a_regs[p] = v;
        }
    };
    load_a(0);

    auto store_rows = [&](float* T, const float4 (&regs)[8]) {  // [row][k], two aligned 8-byte stores per float4
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float* d = T + (p * 4 + r_row) * LDK + r_kq;
            *reinterpret_cast<float2*>(d) = make_float2(regs[p].x, regs[p].y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(regs[p].z, regs[p].w);
        }
    };
    auto store_b_kn = [&]() {  // source [k][n] -> LDS [n][k]
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int k = p * 32 + b_k;
            Bs[(b_nq + 0) * LDK + k] = b_regs[p].x;
            Bs[(b_nq + 1) * LDK + k] = b_regs[p].y;
            Bs[(b_nq + 2) * LDK + k] = b_regs[p].z;
            Bs[(b_nq + 3) * LDK + k] = b_regs[p].w;
        }
    };

    const int wm = (wave >> 1) * 16, wn = (wave & 1) * 16;
    const int fr = lane & 15, fk = lane >> 4;  // fragment row (A) / column (B), and which of the 4 k's of a step
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int np = (g.K + KP - 1) / KP;
    for (int p = 0; p < np; ++p) {
        store_rows(As, a_regs);
        if (BT) store_rows(Bs, b_regs);
        else store_b_kn();
        __syncthreads();
        if (p + 1 < np) {  // the next panel travels while this one is multiplied
            load_b((p + 1) * KP);
            load_a((p + 1) * KP);
        }
        const int klen = min(KP, g.K - p * KP);  // (k beyond it is zero in LDS; a multiple of 4: K % 4 == 0)
        const float* ap = As + (wm + fr) * LDK + fk;
        const float* bp = Bs + (wn + fr) * LDK + fk;
#pragma unroll 8
        for (int kk = 0; kk < klen; kk += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk], bp[kk], acc, 0, 0, 0);
        __syncthreads();
    }

    // ---- epilogue: register r of lane l is tile element (row = 4 * (l >> 4) + r, col = l & 15) ----
    if (MODE == kPlain && g.row_best != nullptr) {
        // argmax epilogue (decoder logits): butterfly over the 16 lanes of a row, the two waves sharing the rows meet
        // in LDS, one (max, lowest index of the max) pair per (row, 32-column tile) goes to memory
        float* best_v = As;                                  // [PM][2]
        int* best_i = reinterpret_cast<int*>(As + PM * 2);   // [PM][2]
        const int n = n0 + wn + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool takes_part = n < g.N && acc[r] > -3.402823466e+38f;  // as decoder_argmax_kernel: NaN / -inf never win
            float v = takes_part ? acc[r] : -3.402823466e+38f;
            int ix = takes_part ? n : -1;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const float ov = __shfl_xor(v, off, kWave);
                const int oi = __shfl_xor(ix, off, kWave);
                const bool take = ov > v || (ov == v && (unsigned)oi < (unsigned)ix);
                v = take ? ov : v;
                ix = take ? oi : ix;
            }
            if (fr == 0) {
                const int row = wm + 4 * fk + r;
                best_v[row * 2 + (wave & 1)] = v;
                best_i[row * 2 + (wave & 1)] = ix;
            }
        }
        __syncthreads();
        if (tid < PM && m0 + tid < g.M) {
            float v = best_v[tid * 2];
            int ix = best_i[tid * 2];
            argmax_take(v, ix, best_v[tid * 2 + 1], best_i[tid * 2 + 1]);
            RowBest* dst = g.row_best + (int64_t)(m0 + tid) * tiles_n + tile;
            *dst = RowBest{v, ix};
        }
        return;
    }
    constexpr bool kCanTranspose = MODE == kNaiveLatest;
    const int64_t o_stride = (kCanTranspose && out_id == 0) ? g.S : 1;  // kt_cache[b, n, s]: element stride S along n
    const int n = n0 + wn + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* op = o_ptr[wm + 4 * fk + r];
        if (op != nullptr && n < g.N) {
            op[(int64_t)n * o_stride] = acc[r];
        }
    }
}

constexpr size_t kPanelSmem = (size_t)(PM + PN) * LDK * sizeof(float) + (size_t)PM * 2 * sizeof(void*);

}  // namespace mli
