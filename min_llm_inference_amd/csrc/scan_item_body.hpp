// The single-pass scan's workgroup body (what it is and why: attention_fused.hip): one (row, chunk) item of the chunked
// grid (fused_decode_scan_kernel), and the publish / merge tail it shares with the contiguous scan (attention_fused_naive.hip).
#pragma once

#include <type_traits>

#include "scan_common.hpp"

namespace mli {

// -DMLI_SCAN_TRACE: every workgroup of the scan records where it ran and when it passed five points (100 MHz
// wall clock), read back by mli_debug_scan_trace -- a diagnostic build for tools/scan_trace.py, never the product.
#ifdef MLI_SCAN_TRACE
constexpr int kTraceSlots = 16384;
__device__ unsigned long long mli_scan_trace[kTraceSlots * 8];
#define MLI_TRACE(i) do { if (threadIdx.x == 0 && trace_id < kTraceSlots) mli_scan_trace[trace_id * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MLI_TRACE(i) do { } while (0)
#endif

constexpr int kFuThreads = 256;
constexpr int kFuWaves = kFuThreads / kWave;

// number of (m, l, partial) triples a row of length L produces: full chunks + pieces of the remainder
__host__ __device__ __forceinline__ int row_items(int L, int ct, int tail) {
    if (tail == 0) return (L + ct - 1) / ct;
    const int nf = L / ct;
    return nf + (L - nf * ct + tail - 1) / tail;
}

// TBR = rows per load batch, MINW = waves per SIMD the register allocator must leave room for
// WAVES = waves per workgroup (each wave owns whole pages; 1 = every wave is its own scheduling unit)
// DS = false: a wave owns whole pages (rows of up to NJ * 64 lane loads) and the waves are merged at the end;
// DS = true ("D-split", wide rows): every wave visits every page of the chunk but owns a slice of NJ * 64 lane loads
//      of each row; the 16 partial scores of a page are exchanged through LDS (one barrier per page, double
//      buffered), after which all waves hold identical softmax state and accumulate their own slice of the output
//      -- perfect balance between the waves however few pages a row has, and no end-of-kernel merge.
// SCORES = true: the reference's contract for the composition -- raw scores go to qkt_output (the combine kernel or
//      the direct path turns them into probabilities with a zero tail).  SCORES = false ("lean" mode, what the layers
//      and engines run): qkt_output is never touched, the only outputs are attention_result and, between the two
//      launches, the per-chunk (max, sum, partial output) triples.
// arrivals != nullptr (lean mode, more than one chunk per row): no combine launch.  Every workgroup publishes its
//      triple write-through (sc1 stores, drained, then one agent-scope add on the row's arrival counter); the
//      workgroup whose add completes the row merges the row's triples in chunk order -- the same expressions in the
//      same order as fused_decode_combine_kernel, so the result is bit-identical to the two-launch form -- and puts
//      the counter back to zero for the next launch.  (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload +
//      every storing wave's vmcnt(0) + barrier + counter add; consumer: agent acquire + vmcnt(0) + barrier, then loads.)
// The tail of every item of a row that has several: the item's partial output row has been stored write-through (sc1)
// at partial_row[c]; here its (m, l) pair follows, the arrival is counted, and the workgroup whose arrival completes the
// row merges the row's nc triples in item order into out_row (MI355X_MICROARCH.md, inter-workgroup visibility: sc1
// payload + every storing wave's vmcnt(0) + barrier + counter add; the last arriver: agent acquire + vmcnt(0) + barrier,
// then loads) and puts the counter back to zero.  All THREADS threads call it.  lds: >= nc float2 of scratch no thread
// still reads; last_sh: one LDS int.
template <int THREADS>
__device__ __forceinline__ void row_publish_merge(float m, float l, float2* ml_row, int c, int nc, unsigned* arrival,
                                                  const float* partial_row, int D, float* out_row, float* lds,
                                                  int* last_sh) {
    typedef unsigned long long __attribute__((address_space(1)))* gu64_ptr;
    typedef unsigned __attribute__((address_space(1)))* gu32_ptr;
    if (threadIdx.x == 0) {
        const unsigned long long packed = ((unsigned long long)__float_as_uint(l) << 32) | __float_as_uint(m);
        __hip_atomic_store((gu64_ptr)(ml_row + c), packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 store
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add((gu32_ptr)arrival, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = before + 1u == (unsigned)nc;
        if (last) {
            __hip_atomic_store((gu32_ptr)arrival, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's L1 copies of the other items' lines
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... and hold the barrier until that has happened
        }
        *last_sh = last;
    }
    __syncthreads();
    if (!*last_sh) return;
    // item statistics -> LDS; sc1 loads: served by L2 / memory, never L1
    float2* ml_sh = reinterpret_cast<float2*>(lds);
    for (int i = threadIdx.x; i < nc; i += THREADS) {
        const unsigned long long packed = __hip_atomic_load((gu64_ptr)(ml_row + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ml_sh[i] = make_float2(__uint_as_float((unsigned)packed), __uint_as_float((unsigned)(packed >> 32)));
    }
    __syncthreads();
    float mm = -INFINITY;
    for (int i = 0; i < nc; ++i) mm = fmaxf(mm, ml_sh[i].x);
    float ll = 0.f;
    for (int i = 0; i < nc; ++i) ll = fmaf(ml_sh[i].y, expf(ml_sh[i].x - mm), ll);
    const float inv_l = 1.f / ll;
    for (int d = 4 * threadIdx.x; d < D; d += 4 * THREADS) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = 0; i0 < nc; i0 += 8) {   // up to 8 partial rows in flight
            fu_u32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (i0 + k < nc) {
                    const float* row_i = partial_row + (int64_t)(i0 + k) * D;
                    const __amdgpu_buffer_rsrc_t prow =
                        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row_i), 0, D * (int)sizeof(float), 0x00020000);
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(prow, d * (int)sizeof(float), 0, 16);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (i0 + k < nc) {
                    const float w = expf(ml_sh[i0 + k].x - mm);
                    r[0] = fmaf(__uint_as_float(v[k].x), w, r[0]);
                    r[1] = fmaf(__uint_as_float(v[k].y), w, r[1]);
                    r[2] = fmaf(__uint_as_float(v[k].z), w, r[2]);
                    r[3] = fmaf(__uint_as_float(v[k].w), w, r[3]);
                }
            }
        }
        *reinterpret_cast<float4*>(out_row + d) = make_float4(r[0] * inv_l, r[1] * inv_l, r[2] * inv_l, r[3] * inv_l);
    }
}

// b, c: the item (row, grid row); first_grid_row: this workgroup is the one that writes the zero result of an empty row.
// RPI = token slots per load instruction (scan_common.hpp; 1 except for narrow fp8 rows): a batch is TBR instructions =
//   TBR * RPI slots
template <class E, int NJ, bool NT, int TBR, int WAVES, bool DS, bool SCORES, int RPI = 1>
__device__ __forceinline__ void fused_scan_item(
    const float* __restrict__ q, const void* const* __restrict__ page_table, const int* __restrict__ lengths,
    float* __restrict__ qkt, float* __restrict__ out, float2* ml, float* partial,
    int S, int D, int ct, int ml_per_row, int nchunk_max, int direct, int tail,
    int slots, unsigned* arrivals, int b, int c, bool first_grid_row, int trace_stride, unsigned char* smem_raw) {
    constexpr int EPL = E::EPL;
    constexpr int LPR = kWave / RPI;   // lanes per token row
    static_assert(RPI == 1 || (NJ == 1 && !DS), "several rows per instruction: rows of one lane load, whole pages per wave");
    const void** ptr_sh = reinterpret_cast<const void**>(smem_raw);                       // ct/16 page pointers
    float* red = reinterpret_cast<float*>(smem_raw + (size_t)(ct / kPage) * 8);            // [waves][NJ*64*EPL]
    __shared__ float2 wave_ml[WAVES];

#ifdef MLI_SCAN_TRACE
    const int trace_id = b + trace_stride * c;
    if (threadIdx.x == 0 && trace_id < kTraceSlots) {
        mli_scan_trace[trace_id * 8 + 5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        mli_scan_trace[trace_id * 8 + 6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
        mli_scan_trace[trace_id * 8 + 1] = 0;
    }
#endif
    MLI_TRACE(0);
    // Prologue chain: lengths[b] -> page pointers -> first K rows, each hop a memory round trip during which this
    // workgroup's share of the CU streams nothing.  Where the item's first token does not depend on the length (the
    // full-chunk grid rows) the pointers -- and q -- are requested BEFORE the length is waited for: one hop less.
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int W = S / kPage;
    const bool early = !tail || c < nchunk_max;
    const void* early_ptr = nullptr;
    if (early && (int)threadIdx.x < ct / kPage && c * (ct / kPage) + (int)threadIdx.x < W)
        early_ptr = page_table[(int64_t)b * W + c * (ct / kPage) + threadIdx.x];
    constexpr int EPLc = E::EPL;
    const int Du = D / EPLc;  // lane-units per row
    // q in registers (NJ * EPL floats per lane), zero beyond the row
    float qr[NJ][EPLc];
    bool live[NJ];
    unsigned voff[NJ];
    const int lane_u = lane % LPR, lane_grp = lane / LPR;   // unit inside the row, row inside the load instruction
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int u = (DS ? wave * NJ * kWave : 0) + lane_u + j * kWave;
        live[j] = u < Du;
        // lanes beyond the row get an offset outside the page block: the buffer range check returns zeros for
        // them, so the loads need no per-lane predication
        voff[j] = live[j] ? (unsigned)u * 16u + (unsigned)lane_grp * (unsigned)(3 * D * E::kBytes) : 0x40000000u;
#pragma unroll
        for (int e = 0; e < EPLc; ++e) qr[j][e] = live[j] ? q[(int64_t)b * D + u * EPLc + e] : 0.f;
    }
    const int L = min(lengths[b], S);
    if (arrivals != nullptr && L == 0) {
        // in-kernel merge: no workgroup arrives for an empty row, so its zero result is written here, once
        if (first_grid_row) {
            for (int i = threadIdx.x; i < D; i += (WAVES * kWave)) out[(int64_t)b * D + i] = 0.f;
        }
        return;
    }
    // Items of a row, in token order: its full chunks, then (tail > 0) the remainder cut into pieces of `tail` tokens.
    // Largest items first: grid rows 0 .. nchunk-1 run only the FULL chunks, the grid rows behind them the pieces of
    // every row's remainder.  In plain chunk order the last workgroups to start are often full ones and the launch
    // ends with a long stretch at a fraction of the bandwidth (tools/scan_trace.py); with the pieces last, what is
    // still running when the queue runs dry is at most `tail` tokens long.
    int s0 = c * ct, s1 = min(s0 + ct, L);
    if (tail) {
        const int nf = L / ct;
        if (c < nchunk_max) {
            if (c >= nf) return;                // empty, or part of the remainder (the grid rows behind take it)
        } else {
            s0 = nf * ct + (c - nchunk_max) * tail;
            if (s0 >= L) return;                // (covers the empty row)
            s1 = min(s0 + tail, L);
            c = nf + (c - nchunk_max);          // its slot among the row's items
        }
    }
    float* qkt_row = qkt + (int64_t)b * S;

    if (s0 >= L) {
        if (direct) {  // single-chunk problem: this workgroup owns the whole (empty) row
            if (SCORES) for (int i = threadIdx.x; i < S; i += (WAVES * kWave)) qkt_row[i] = 0.f;
            for (int i = threadIdx.x; i < D; i += (WAVES * kWave)) out[(int64_t)b * D + i] = 0.f;
        }
        return;
    }
    const int ntok = s1 - s0;
    const int npages = (ntok + kPage - 1) / kPage;
    if (early) {
        if ((int)threadIdx.x < npages) ptr_sh[threadIdx.x] = early_ptr;   // npages <= ct / 16 <= 64 < threads
    } else {
        for (int i = threadIdx.x; i < npages; i += (WAVES * kWave))
            ptr_sh[i] = page_table[(int64_t)b * W + s0 / kPage + i];
    }
    __syncthreads();
    MLI_TRACE(1);

    const float scale = sqrtf((float)D);
    const int64_t row_bytes = (int64_t)3 * D * E::kBytes;  // consecutive token slots of a page
    const int64_t seg_bytes = (int64_t)D * E::kBytes;      // segment stride inside a slot: x | K | V

    float run_m = -INFINITY, run_l = 0.f;
    float acc[NJ][EPL];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[j][e] = 0.f;
    // Rolling prefetch over row batches.  A page is 2 * NB batches of TBR rows (K batches, then V batches); batch
    // `pos` of every page lives in register buffer pos % 4, and before batch `pos` is consumed batch pos + 3 -- of
    // this page or of the wave's next page -- is issued, so three batches (24 KiB at bf16 D=512) stay in flight
    // per wave across the butterfly reduction, the softmax update and the page boundary.
    constexpr int NB = 16 / (TBR * RPI);
    constexpr int NPOS = 2 * NB;
    constexpr int PD = 3;
    fu_u32x4 buf[4][TBR][NJ];

    // Loads go through a buffer descriptor built from the wave-uniform page pointer: the 128-bit descriptor and
    // the per-row offset live in SGPRs, each lane contributes one 32-bit byte offset (no 64-bit per-load address
    // VGPRs), and the hardware range check (one page block) backs up the indexing.
    const int block_bytes = kPage * 3 * D * E::kBytes;
    auto page_ptr = [&](int pi) {
        return reinterpret_cast<const char*>(wave_uniform(reinterpret_cast<const float*>(ptr_sh[pi])));
    };
    auto issue = [&](auto POS, const char* pg) {
        constexpr int pos = decltype(POS)::value;
        constexpr int bi = pos % 4;
        // re-assert uniformity at the point of use: `pg` went through selects on the wave index, which the compiler
        // treats as divergent and would wrap every load in a waterfall loop
        const char* upg = reinterpret_cast<const char*>(wave_uniform(reinterpret_cast<const float*>(pg)));
        // a null page (row longer than its pages: a caller bug) gets an empty range: its loads return zeros
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(upg), 0, upg != nullptr ? block_bytes : 0, 0x00020000);
        const int base = (pos < NB ? (int)seg_bytes : 2 * (int)seg_bytes) + (pos % NB) * TBR * RPI * (int)row_bytes;
#pragma unroll
        for (int t = 0; t < TBR; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                buf[bi][t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], base + t * RPI * (int)row_bytes, NT ? 2 : 0);
    };

    constexpr int PSTEP = DS ? 1 : WAVES;
    const int p_first = DS ? 0 : wave;
    const char* page = p_first < npages ? page_ptr(p_first) : nullptr;
    if (p_first < npages) {   // (not "page != nullptr": a null table entry is a page too -- it reads as zeros)
        issue(std::integral_constant<int, 0>{}, page);
        issue(std::integral_constant<int, 1>{}, page);
        issue(std::integral_constant<int, 2>{}, page);
    }
    for (int pi = p_first; pi < npages; pi += PSTEP) {
        const bool has_next = pi + PSTEP < npages;
        const char* next = has_next ? page_ptr(pi + PSTEP) : nullptr;
        const int nt = min(kPage, ntok - pi * kPage);  // live tokens in this page (>= 1)
        float sacc[16 / RPI];
#pragma unroll
        for (int t = 0; t < 16 / RPI; ++t) sacc[t] = 0.f;
        float p_lane = 0.f;

        static_for<NPOS>([&](auto POS) {
            constexpr int pos = decltype(POS)::value;
            constexpr int bi = pos % 4;
            constexpr int tgt = pos + PD;
            if constexpr (tgt < NPOS) {
                issue(std::integral_constant<int, tgt>{}, page);
            } else {
                if (has_next) issue(std::integral_constant<int, tgt - NPOS>{}, next);  // wave-uniform
            }
            if constexpr (pos < NB) {
                // ---- K batch: partial scores of slots pos*TBR .. pos*TBR+TBR-1 ----
#pragma unroll
                for (int t = 0; t < TBR; ++t)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) ElemMath<E>::dot(buf[bi][t][j], qr[j], sacc[pos * TBR + t]);
                if constexpr (pos == NB - 1) {
                    // all 16 slots scored (slots >= nt hold allocated but meaningless data: masked here)
                    float tot = rpi_reduce<RPI>(sacc, lane);  // lane holds the sum for slot (lane >> 2) & 15 (RPI = 1)
                    const int slot = rpi_slot_of_lane<RPI>(lane);
                    if constexpr (DS) {
                        // complete the dot products across the waves' row slices (fixed order: identical in every wave)
                        float* xs = red + (pi & 1) * (WAVES * 16);
                        if ((lane & 3) == 0) xs[wave * 16 + slot] = tot;
                        __syncthreads();
                        tot = 0.f;
#pragma unroll
                        for (int w = 0; w < WAVES; ++w) tot += xs[w * 16 + slot];
                    }
                    const bool valid = slot < nt;
                    const float score = tot / scale;
                    if (SCORES && valid && (lane & 3) == 0 && (!DS || wave == 0))
                        qkt_row[s0 + pi * kPage + slot] = score;  // raw; normalised later
                    // online softmax update
                    const float pm = wave_max(valid ? score : -INFINITY);
                    const float m_new = fmaxf(run_m, pm);
                    const float alpha = run_m == -INFINITY ? 0.f : expf(run_m - m_new);
                    p_lane = valid ? expf(score - m_new) : 0.f;
                    run_l = run_l * alpha + wave_sum((lane & 3) == 0 ? p_lane : 0.f);
                    run_m = m_new;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int e = 0; e < EPL; ++e) acc[j][e] *= alpha;
                }
            } else {
                // ---- V batch: acc += p . V over the live slots ----
                constexpr int first = (pos - NB) * TBR;
#pragma unroll
                for (int t = 0; t < TBR; ++t) {
                    // the slot's probability sits in lane 4 * slot: broadcast through an SGPR
                    const float p = rpi_prob<RPI>(p_lane, first + t, lane);
                    if (RPI * (first + t) < nt) {  // wave-uniform: never multiply unwritten page memory, even by zero
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            fu_u32x4 raw = buf[bi][t][j];
                            if constexpr (RPI > 1) {   // (per lane group: a slot beyond the row reads as zeros)
                                const bool ok = RPI * (first + t) + lane_grp < nt;
                                raw.x = ok ? raw.x : 0u; raw.y = ok ? raw.y : 0u; raw.z = ok ? raw.z : 0u; raw.w = ok ? raw.w : 0u;
                            }
                            ElemMath<E>::axpy(raw, p, acc[j]);
                        }
                    }
                }
            }
        });
        page = next;
#ifdef MLI_SCAN_TRACE
        if (pi == p_first) MLI_TRACE(2);
#endif
    }
    MLI_TRACE(3);

    // ---- the chunk's result: (m, l) and the un-normalised partial output row ----
    // DS: every wave already holds the chunk's (max, sum) and its own slice of the output.
    // otherwise: the waves (each owns whole pages) are merged in wave order through LDS.
    const bool publish = arrivals != nullptr;  // lean mode, several chunks per row: in-kernel merge by the last arriver
    float* o = direct ? out + (int64_t)b * D : partial + ((int64_t)b * slots + c) * D;
    float m, l;
    if constexpr (DS) {
        m = run_m;
        l = run_l;
        const float norm = direct ? 1.f / run_l : 1.f;
        // write-through (sc1) stores when another workgroup will read the row back inside this launch
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(o, 0, D * (int)sizeof(float), 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (!live[j]) continue;
            const int u = wave * NJ * kWave + lane + j * kWave;
#pragma unroll
            for (int e = 0; e < EPL; e += 4) {
                const float4 v = make_float4(acc[j][e] * norm, acc[j][e + 1] * norm, acc[j][e + 2] * norm, acc[j][e + 3] * norm);
                if (publish) {
                    fu_u32x4 raw;
                    raw.x = __float_as_uint(v.x); raw.y = __float_as_uint(v.y); raw.z = __float_as_uint(v.z); raw.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, orow, (u * EPL + e) * (int)sizeof(float), 0, 16);
                } else {
                    *reinterpret_cast<float4*>(o + (int64_t)u * EPL + e) = v;
                }
            }
        }
    } else {
        constexpr int kRowF = NJ * LPR * EPL;  // floats one wave contributes
        if (lane == 0) wave_ml[wave] = make_float2(run_m, run_l);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float a = rpi_group_sum<RPI>(acc[j][e]);   // the lane groups hold different slots' contributions
                if (RPI == 1 || lane < LPR) red[wave * kRowF + (j * kWave + lane_u) * EPL + e] = a;
            }
        __syncthreads();
        m = -INFINITY;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) m = fmaxf(m, wave_ml[w].x);
        float wsc[WAVES];
        l = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            wsc[w] = wave_ml[w].x == -INFINITY ? 0.f : expf(wave_ml[w].x - m);
            l += wave_ml[w].y * wsc[w];
        }
        const float norm = direct ? 1.f / l : 1.f;
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(o, 0, D * (int)sizeof(float), 0x00020000);
        // element i of the row lives at red[...][i] by construction; D % 4 == 0
        for (int i = 4 * threadIdx.x; i < D; i += 4 * (WAVES * kWave)) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) t += red[w * kRowF + i + k] * wsc[w];
                r[k] = t * norm;
            }
            if (publish) {
                fu_u32x4 raw;
                raw.x = __float_as_uint(r[0]); raw.y = __float_as_uint(r[1]); raw.z = __float_as_uint(r[2]); raw.w = __float_as_uint(r[3]);
                __builtin_amdgcn_raw_buffer_store_b128(raw, orow, i * (int)sizeof(float), 0, 16);
            } else {
                *reinterpret_cast<float4*>(o + i) = make_float4(r[0], r[1], r[2], r[3]);
            }
        }
    }
    if (direct) {
        if (SCORES) {
            // whole row handled by this workgroup: normalise the scores in place and write the zero tail
            __syncthreads();  // raw scores written by other waves of this workgroup are visible after the barrier
            const float inv_l = 1.f / l;
            for (int i = threadIdx.x; i < S; i += (WAVES * kWave)) qkt_row[i] = i < L ? expf(qkt_row[i] - m) * inv_l : 0.f;
        }
    } else if (!publish) {
        if (threadIdx.x == 0) ml[(int64_t)b * ml_per_row + c] = make_float2(m, l);
    } else {
        // ---- publish the triple, count the arrival; the workgroup that completes the row merges it ----
        row_publish_merge<WAVES * kWave>(m, l, ml + (int64_t)b * ml_per_row, c, row_items(L, ct, tail), arrivals + b,
                                         partial + (int64_t)b * slots * D, D, out + (int64_t)b * D, red,
                                         reinterpret_cast<int*>(wave_ml));
    }
    MLI_TRACE(4);
#ifdef MLI_SCAN_TRACE
    if (threadIdx.x == 0 && trace_id < kTraceSlots) mli_scan_trace[trace_id * 8 + 7] = (unsigned long long)npages;
#endif
}

}  // namespace mli
