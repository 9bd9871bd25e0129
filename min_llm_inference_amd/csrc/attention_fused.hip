// Single-pass decode attention over paged KV ("flash-decoding" form) used by the paged_attention compositions:
// every page is visited ONCE and both its K rows and its V rows are consumed in that visit, with an online
// (running max / running sum) softmax per wave.  Compared with the separate q.K^T and softmax.V passes this halves
// the number of page visits (one TLB / DRAM-page walk per 2/3 of a block instead of per 1/3 -- pages are scattered
// over the pool), removes the softmax launch and keeps the scores out of the critical path.
//
// The reference's contract for the composition (src/kernels/paged_attention.cu:358-377) is kept: on return
// qkt_output holds the masked softmax probabilities with a zero tail, attention_result holds sum p.V, rows with
// length 0 give zeros.  Raw scores are written by the scan kernel; the combine kernel merges the per-chunk
// (max, sum, partial output) triples in chunk order and normalises the row.
//
//   scan    grid = (B, chunks) rows fast (XCD balance), 256 threads; a wave owns whole pages
//   combine grid = B, 256 threads
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

int sv_chunk_tokens_for(int n_batch, int n_sequence);  // attention_scan.hip
int tuned_chunk_tokens();
template <class E>
int launch_stream_decode(const float* q, const void* const* page_table, const int* lengths, float* out, int B, int S, int D,
                         void* ws, size_t ws_bytes, hipStream_t st);   // attention_stream.hip
template <class E>
bool stream_decode_applies(int B, int S, int D);
size_t stats_region_bytes_for(int B, int S);
int nt_loads_for(int B, int S, int D, int esize);

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
template <class E, int NJ, bool NT, int TBR, int MINW, int WAVES, bool DS = false, bool SCORES = true>
__global__ __launch_bounds__(WAVES * kWave, MINW) void fused_decode_scan_kernel(
    const float* __restrict__ q, const void* const* __restrict__ page_table, const int* __restrict__ lengths,
    float* __restrict__ qkt, float* __restrict__ out, float2* ml, float* partial,
    int S, int D, int ct, int ml_per_row, int nchunk_max, int direct, unsigned* __restrict__ ticket, int tail,
    int slots, unsigned* arrivals) {
    constexpr int EPL = E::EPL;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const void** ptr_sh = reinterpret_cast<const void**>(smem_raw);                       // ct/16 page pointers
    float* red = reinterpret_cast<float*>(smem_raw + (size_t)(ct / kPage) * 8);            // [waves][NJ*64*EPL]
    __shared__ float2 wave_ml[WAVES];

    // Which (row, chunk) this workgroup takes.  Static: its grid position.  With a ticket counter: the next item in
    // the same order (rows fast, then chunks), whichever workgroup asks first.  Workgroups are dealt to the 8 XCDs
    // round-robin by grid position, so the static form gives every XCD a fixed eighth of the rows -- with ragged
    // lengths the XCDs' totals differ by +-10 % and the slowest one sets the kernel time (tools/scan_trace.py: last
    // workgroup of an XCD at 615..701 us); tickets let the XCDs that run ahead take more items.
    // Workgroups are dealt to the 8 XCDs round-robin by linear grid position, i.e. by blockIdx.x % 8 (n_batch is a
    // multiple of 8 in every configuration that fills the chip).  Rotating the row index by the chunk index spreads
    // the chunks of one row over the XCDs, so every XCD streams a mix of all rows instead of a fixed eighth of them
    // whose total length differs from the others' by +-10 % (tools/scan_trace.py: XCDs done at 615 .. 701 us).
    int c = blockIdx.y;
    int b = (int)((blockIdx.x + (unsigned)c) % gridDim.x);
    if (ticket != nullptr) {
        __shared__ unsigned item_sh;
        if (threadIdx.x == 0) item_sh = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned item = item_sh;
        if (item >= gridDim.x * gridDim.y) return;  // cannot happen with a zeroed counter; never index out of range
        b = (int)(item % gridDim.x);
        c = (int)(item / gridDim.x);
    }
#ifdef MLI_SCAN_TRACE
    const int trace_id = b + gridDim.x * c;
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
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int u = (DS ? wave * NJ * kWave : 0) + lane + j * kWave;
        live[j] = u < Du;
        // lanes beyond the row get an offset outside the page block: the buffer range check returns zeros for
        // them, so the loads need no per-lane predication
        voff[j] = live[j] ? (unsigned)u * 16u : 0x40000000u;
#pragma unroll
        for (int e = 0; e < EPLc; ++e) qr[j][e] = live[j] ? q[(int64_t)b * D + u * EPLc + e] : 0.f;
    }
    const int L = min(lengths[b], S);
    if (arrivals != nullptr && L == 0) {
        // in-kernel merge: no workgroup arrives for an empty row, so its zero result is written here, once
        if (blockIdx.y == 0) for (int i = threadIdx.x; i < D; i += (WAVES * kWave)) out[(int64_t)b * D + i] = 0.f;
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
    constexpr int NB = 16 / TBR;
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
        const int base = (pos < NB ? (int)seg_bytes : 2 * (int)seg_bytes) + (pos % NB) * TBR * (int)row_bytes;
#pragma unroll
        for (int t = 0; t < TBR; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                buf[bi][t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], base + t * (int)row_bytes, NT ? 2 : 0);
    };

    constexpr int PSTEP = DS ? 1 : WAVES;
    const int p_first = DS ? 0 : wave;
    const char* page = p_first < npages ? page_ptr(p_first) : nullptr;
    if (p_first < npages) {  // (not "page != nullptr": a null table entry is a page too -- it reads as zeros)
        issue(std::integral_constant<int, 0>{}, page);
        issue(std::integral_constant<int, 1>{}, page);
        issue(std::integral_constant<int, 2>{}, page);
    }
    for (int pi = p_first; pi < npages; pi += PSTEP) {
        const bool has_next = pi + PSTEP < npages;
        const char* next = has_next ? page_ptr(pi + PSTEP) : nullptr;
        const int nt = min(kPage, ntok - pi * kPage);  // live tokens in this page (>= 1)
        float sacc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) sacc[t] = 0.f;
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
                    for (int j = 0; j < NJ; ++j) {
                        float kf[EPL];
                        E::unpack(buf[bi][t][j], kf);
#pragma unroll
                        for (int e = 0; e < EPL; ++e) sacc[pos * TBR + t] = fmaf(qr[j][e], kf[e], sacc[pos * TBR + t]);
                    }
                if constexpr (pos == NB - 1) {
                    // all 16 slots scored (slots >= nt hold allocated but meaningless data: masked here)
                    float tot = wave_reduce16(sacc, lane);  // lane holds the sum for slot (lane >> 2) & 15
                    const int slot = (lane >> 2) & 15;
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
                    const float p = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(p_lane), 4 * (first + t)));
                    if (first + t < nt) {  // wave-uniform: never multiply unwritten page memory, even by zero
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            float vf[EPL];
                            E::unpack(buf[bi][t][j], vf);
#pragma unroll
                            for (int e = 0; e < EPL; ++e) acc[j][e] = fmaf(p, vf[e], acc[j][e]);
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
        constexpr int kRowF = NJ * kWave * EPL;  // floats one wave contributes
        if (lane == 0) wave_ml[wave] = make_float2(run_m, run_l);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < EPL; ++e) red[wave * kRowF + (j * kWave + lane) * EPL + e] = acc[j][e];
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
        typedef unsigned long long __attribute__((address_space(1)))* gu64_ptr;
        typedef unsigned __attribute__((address_space(1)))* gu32_ptr;
        int* last_sh = reinterpret_cast<int*>(wave_ml);  // free by now: every read of wave_ml is behind a barrier
        float2* ml_row = ml + (int64_t)b * ml_per_row;
        if (threadIdx.x == 0) {
            const unsigned long long packed = ((unsigned long long)__float_as_uint(l) << 32) | __float_as_uint(m);
            __hip_atomic_store((gu64_ptr)(ml_row + c), packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1 store
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave: its write-through stores have left
        __syncthreads();
        const int nc = row_items(L, ct, tail);            // items of this row that do work, i.e. arrivals to expect
        if (threadIdx.x == 0) {
            const unsigned before = __hip_atomic_fetch_add((gu32_ptr)(arrivals + b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = before + 1u == (unsigned)nc;
            if (last) {
                __hip_atomic_store((gu32_ptr)(arrivals + b), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's L1 copies of the other chunks' lines
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... and hold the barrier until that has happened
            }
            *last_sh = last;
        }
        __syncthreads();
        if (*last_sh) {
            // chunk statistics -> LDS (the scan's reduction buffer is free now); sc1 loads: served by L2 / memory, never L1
            float2* ml_sh = reinterpret_cast<float2*>(red);
            for (int i = threadIdx.x; i < nc; i += (WAVES * kWave)) {
                const unsigned long long packed = __hip_atomic_load((gu64_ptr)(ml_row + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ml_sh[i] = make_float2(__uint_as_float((unsigned)packed), __uint_as_float((unsigned)(packed >> 32)));
            }
            __syncthreads();
            float mm = -INFINITY;
            for (int i = 0; i < nc; ++i) mm = fmaxf(mm, ml_sh[i].x);
            float ll = 0.f;
            for (int i = 0; i < nc; ++i) ll = fmaf(ml_sh[i].y, expf(ml_sh[i].x - mm), ll);
            const float inv_l = 1.f / ll;
            const float* pr = partial + (int64_t)b * slots * D;
            for (int d = 4 * threadIdx.x; d < D; d += 4 * (WAVES * kWave)) {
                float r[4] = {0.f, 0.f, 0.f, 0.f};
                for (int i0 = 0; i0 < nc; i0 += 8) {   // up to 8 chunk rows in flight
                    fu_u32x4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (i0 + k < nc) {
                            const float* row_i = pr + (int64_t)(i0 + k) * D;
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
                *reinterpret_cast<float4*>(out + (int64_t)b * D + d) =
                    make_float4(r[0] * inv_l, r[1] * inv_l, r[2] * inv_l, r[3] * inv_l);
            }
        }
    }
    MLI_TRACE(4);
#ifdef MLI_SCAN_TRACE
    if (threadIdx.x == 0 && trace_id < kTraceSlots) mli_scan_trace[trace_id * 8 + 7] = (unsigned long long)npages;
#endif
}

// grid = (B, kCombineParts).  Every part merges the row's chunk statistics (cheap, identical result), part 0 also
// writes attention_result; each part turns its slice of the raw scores into probabilities (zero tail included).
constexpr int kCombineParts = 4;

__global__ __launch_bounds__(kFuThreads) void fused_decode_combine_kernel(
    const float2* __restrict__ ml, const float* __restrict__ partial, const int* __restrict__ lengths,
    float* __restrict__ qkt, float* __restrict__ out, int S, int D, int ct, int ml_per_row, int slots, int tail) {
    const int b = blockIdx.x;
    const int part = blockIdx.y;
    const int L = min(lengths[b], S);
    const int nc = row_items(L, ct, tail);
    float* qkt_row = qkt + (int64_t)b * S;
    const int per = ((S + kCombineParts - 1) / kCombineParts + 3) & ~3;
    const int i0 = part * per, i1 = min(S, i0 + per);
    if (nc == 0) {
        if (qkt != nullptr) for (int i = i0 + threadIdx.x; i < i1; i += kFuThreads) qkt_row[i] = 0.f;
        if (part == 0) for (int i = threadIdx.x; i < D; i += kFuThreads) out[(int64_t)b * D + i] = 0.f;
        return;
    }
    const float2* row = ml + (int64_t)b * ml_per_row;
    float m = -INFINITY;
    for (int i = 0; i < nc; ++i) m = fmaxf(m, row[i].x);
    float l = 0.f;
    for (int i = 0; i < nc; ++i) l = fmaf(row[i].y, expf(row[i].x - m), l);
    const float inv_l = 1.f / l;
    if (part == 0) {
        const float* pr = partial + (int64_t)b * slots * D;
        for (int d = threadIdx.x; d < D; d += kFuThreads) {
            float r = 0.f;
            for (int i = 0; i < nc; ++i) r = fmaf(pr[(int64_t)i * D + d], expf(row[i].x - m), r);
            out[(int64_t)b * D + d] = r * inv_l;
        }
    }
    if (qkt == nullptr) return;  // lean mode: the scores were never written
    for (int i = i0 + threadIdx.x; i < i1; i += kFuThreads) qkt_row[i] = i < L ? expf(qkt_row[i] - m) * inv_l : 0.f;
}

static int g_flash = 1;
static int g_flash_variant = 0;  // register-budget variants of the scan kernel (tuning)
// mli_tune "scan_partial_last": 1 (default) = full chunks first, every row's remainder behind them in pieces of
// "scan_tail_tokens" tokens; 0 = plain chunk order
static int g_partial_last = 1;
void set_partial_last(int v) { g_partial_last = v != 0; }
// mli_tune "scan_tail_tokens": 0 (default) = the whole remainder as one piece, else a power of two in [64, chunk].
// Measured at config 4 (bf16, 512-token chunks, lean form): one piece 658.7 us, 256-token pieces 661.9, 128: 668.7,
// 64: 688.0 -- every item costs about 2.5 us of a workgroup slot (prologue chain lengths -> page pointers -> first K
// rows, epilogue merge and publication), more than the shorter end of the launch gives back.
static int g_tail_tokens = 0;
void set_tail_tokens(int v) { g_tail_tokens = v; }
// mli_tune "scan_dynamic_items": ticketed (row, chunk) assignment.  Off by default: it shortens the kernel by 0.6-1.5 %
// (4-10 us at config 4), and the hipMemsetAsync that zeroes the counter before every launch costs the stream ~8 us.
static int g_dynamic_items = 0;
void set_dynamic_items(int v) { g_dynamic_items = v != 0; }
// mli_tune "scan_merge" (lean mode only): 1 (default) = the workgroup that completes a row merges its chunks inside
// the scan launch, 0 = the separate combine launch (bit-identical results)
static int g_scan_merge = 1;
void set_scan_merge(int v) { g_scan_merge = v != 0; }
void set_flash_decode(int v) { g_flash = v != 0; }
void set_flash_variant(int v) { g_flash_variant = v; }

// Tokens per workgroup of the single-pass scan: the largest power of two <= 512 that still cuts the batch into
// >= 2048 (row, chunk) slots, i.e. with ragged lengths about two rounds of real items for the 512 workgroups the chip
// holds.  Measured: B=1024, S=4096 -> 512 (256: +2.4 %, 1024: +1 % with ragged lengths); B=256, S=1024 -> 128 (round 2,
// scan launch: 46.9 us against 49.6 at 256 and 54.3 at 512, where 384 items of very unequal size cannot even fill the
// 512 slots once; lean form 52.5 / 53.1 / 56.2).
static int fused_chunk_tokens(int B, int S) {
    if (tuned_chunk_tokens() != 0) return sv_chunk_tokens_for(B, S);  // forced (mli_tune / MLI_CHUNK_TOKENS)
    int ct = 512;
    while (ct > 64 && (int64_t)B * ceil_div_i(S, ct) < 2048) ct >>= 1;
    return ct;
}

// returns 1 when the fused path ran, 0 when the caller should take the three-kernel path, < 0 / > 1 on error
// phases: bit 0 = scan kernel, bit 1 = combine kernel (3 = the whole block; 1 / 2 let bench.py time them apart),
//         bit 2 = lean mode: qkt is neither read nor written (may be null), and with "scan_merge" on there is no
//         combine launch -- the scan merges each row itself (phases 5 then does the whole job, 6 nothing)
template <class E>
static int launch_fused_decode(const float* q, const void* const* page_table, const int* lengths, float* qkt,
                               float* out, int B, int S, int D, void* ws, size_t ws_bytes, hipStream_t st,
                               int phases = 3) {
    const bool lean = (phases & 4) != 0;
    if (lean && g_scan_merge && g_flash) {
        // chip-filling batches: equal page shares instead of (row, chunk) workgroups (attention_stream.hip); one launch
        // does the whole job, so the "combine only" phase has nothing left to do
        const int r = (phases & 1) ? launch_stream_decode<E>(q, page_table, lengths, out, B, S, D, ws, ws_bytes, st)
                                   : (stream_decode_applies<E>(B, S, D) ? 1 : 0);
        if (r != 0) return r;
    }
    const int Du = D / E::EPL;
    const int nj = ceil_div_i(Du, kWave);
    if (!g_flash || nj > 8 || D % E::EPL != 0 || S % kPage != 0) return 0;
    const bool dsplit = nj > 2;  // wide rows: the four waves split the row instead of the pages
    const int nj_ds = ceil_div_i(Du, kWave * kFuWaves);  // 1 or 2
    // variant 3: single-wave workgroups of 128 tokens -- every wave is its own scheduling unit, no LDS merge,
    // no barrier; the hardware dispatcher does the load balancing
    const bool solo = g_flash_variant == 3 && S > 128 && !dsplit && !(phases & 4);
    // short sequences with a full batch: one workgroup per row (no partials, no combine launch) beats two 64-token
    // chunks (README workload, S = 128: 200 vs 209 us)
    const int ct = solo ? 128 : (S <= 128 && B >= 256 && tuned_chunk_tokens() == 0) ? 128 : fused_chunk_tokens(B, S);
    const int nchunk = ceil_div_i(S, ct);
    const int direct = nchunk == 1;
    const size_t stats_bytes = stats_region_bytes_for(B, S);
    float2* ml = nullptr;
    float* partial = nullptr;
    if (!direct) {
        if (ws == nullptr || ws_bytes < stats_bytes + (size_t)B * nchunk * D * sizeof(float)) return 0;
        ml = reinterpret_cast<float2*>(ws);
        partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + stats_bytes);
    }
    const int ml_per_row = ceil_div_i(S, 64);
    // lean mode: arrival counters (one per row, zero between launches) in front of the workspace body
    unsigned* arrivals = nullptr;
    if (lean && !direct && g_scan_merge && B <= kMaxArrivalRows) arrivals = ws_arrivals(ws);  // ws != nullptr: checked above
    // ticket counter for the dynamic (row, chunk) assignment: in the part of the workspace the partial sums of this
    // chunk size leave unused, zeroed on the stream before every launch
    unsigned* ticket = nullptr;
    if (!direct && g_dynamic_items && (phases & 1) && (int64_t)B * nchunk >= 1024) {
        const size_t off = (stats_bytes + (size_t)B * nchunk * D * sizeof(float) + 255) & ~(size_t)255;
        if (off + sizeof(unsigned) <= ws_bytes) {
            ticket = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + off);
            if (hipMemsetAsync(ticket, 0, sizeof(unsigned), st) != hipSuccess) ticket = nullptr;
        }
    }
    const int waves = solo ? 1 : kFuWaves;
    // page pointers of the chunk | reduction buffer (also holds the row's chunk statistics during the in-kernel merge)
    const size_t red_bytes = (dsplit ? (size_t)2 * kFuWaves * 16 : (size_t)waves * nj * kWave * E::EPL) * sizeof(float);
    const size_t stat_bytes_row = (size_t)ml_per_row * 8;  // upper bound of the triples a row can have
    const size_t smem = (size_t)(ct / kPage) * 8 + (red_bytes > stat_bytes_row ? red_bytes : stat_bytes_row);
    dim3 grid(B, nchunk);
    // tail > 0: full chunks first, the remainders behind them in pieces of `tail` tokens (slots = triples per row)
    int tail = 0, slots = nchunk;
    if (!direct && ticket == nullptr && g_partial_last) {
        tail = g_tail_tokens ? g_tail_tokens : ct;
        if (tail > ct || tail < 64 || (tail & (tail - 1))) tail = ct;
        // a row with a remainder has at most nchunk - 1 full chunks: nchunk + pieces - 1 triples per row at most (the
        // workspace is sized for 64-token chunks: room for every layout with no more items per row than that)
        if (nchunk + ct / tail - 1 > ml_per_row) tail = ct;
        slots = nchunk + ct / tail - 1;
        grid = dim3(B, nchunk + ct / tail);
        if ((size_t)B * slots * D * sizeof(float) + stats_bytes > ws_bytes) return 0;
    }
    const bool nt = nt_loads_for(B, S, D, E::kBytes);
#define MLI_FU_LAUNCH(NJ, NT, TBR, MINW, WAVES, ...)                                                              \
    hipLaunchKernelGGL((fused_decode_scan_kernel<E, NJ, NT, TBR, MINW, WAVES, ##__VA_ARGS__>), grid,               \
                       dim3(WAVES * kWave), smem, st, q, page_table, lengths, qkt, out, ml, partial, S, D, ct,     \
                       ml_per_row, nchunk, direct, ticket, tail, slots, arrivals)
    if ((phases & 1) && lean) {
        // the default register budget only (the variants are tuning experiments of the materialising form)
        if (dsplit) {
            if (nj_ds == 1) {
                if (nt) MLI_FU_LAUNCH(1, true, 8, 2, 4, true, false);
                else MLI_FU_LAUNCH(1, false, 8, 2, 4, true, false);
            } else {
                if (nt) MLI_FU_LAUNCH(2, true, 4, 2, 4, true, false);
                else MLI_FU_LAUNCH(2, false, 4, 2, 4, true, false);
            }
        } else if (nj == 1) {
            if (nt) MLI_FU_LAUNCH(1, true, 8, 2, 4, false, false);
            else MLI_FU_LAUNCH(1, false, 8, 2, 4, false, false);
        } else {
            if (nt) MLI_FU_LAUNCH(2, true, 4, 2, 4, false, false);
            else MLI_FU_LAUNCH(2, false, 4, 2, 4, false, false);
        }
    } else if (phases & 1) {
        if (dsplit) {
            if (nj_ds == 1) {
                if (nt) MLI_FU_LAUNCH(1, true, 8, 2, 4, true);
                else MLI_FU_LAUNCH(1, false, 8, 2, 4, true);
            } else {
                if (nt) MLI_FU_LAUNCH(2, true, 4, 2, 4, true);
                else MLI_FU_LAUNCH(2, false, 4, 2, 4, true);
            }
        } else if (nj == 1) {
            if (!nt) MLI_FU_LAUNCH(1, false, 8, 2, 4);
            else if (solo) MLI_FU_LAUNCH(1, true, 8, 2, 1);
            else if (g_flash_variant == 2) MLI_FU_LAUNCH(1, true, 4, 4, 4);
            else MLI_FU_LAUNCH(1, true, 8, 2, 4);
        } else {
            if (!nt) MLI_FU_LAUNCH(2, false, 4, 2, 4);
            else if (solo) MLI_FU_LAUNCH(2, true, 4, 2, 1);
            else if (g_flash_variant == 2) MLI_FU_LAUNCH(2, true, 2, 4, 4);
            else MLI_FU_LAUNCH(2, true, 4, 2, 4);
        }
    }
#undef MLI_FU_LAUNCH
    int rc = launch_status();
    if (rc) return rc > 0 ? rc + 1 : rc;  // keep 1 free for "ran"
    if (!direct && (phases & 2) && arrivals == nullptr) {
        // lean: one part per row, no score pass (qkt == nullptr)
        hipLaunchKernelGGL(fused_decode_combine_kernel, dim3(B, lean ? 1 : kCombineParts), dim3(kFuThreads), 0, st, ml,
                           partial, lengths, lean ? nullptr : qkt, out, S, D, ct, ml_per_row, slots, tail);
        rc = launch_status();
        if (rc) return rc > 0 ? rc + 1 : rc;
    }
    return 1;
}

// qkt == nullptr selects the lean mode (no scores, in-kernel merge)
int launch_fused_decode_f32(const float* q, const float* const* page_table, const int* lengths, float* qkt, float* out,
                            int B, int S, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    return launch_fused_decode<ElemF32>(q, reinterpret_cast<const void* const*>(page_table), lengths, qkt, out, B, S, D,
                                        ws, ws_bytes, st, qkt == nullptr ? 7 : 3);
}

int launch_fused_decode_bf16(const float* q, const uint16_t* const* page_table, const int* lengths, float* qkt,
                             float* out, int B, int S, int D, void* ws, size_t ws_bytes, hipStream_t st) {
    return launch_fused_decode<ElemBF16>(q, reinterpret_cast<const void* const*>(page_table), lengths, qkt, out, B, S, D,
                                         ws, ws_bytes, st, qkt == nullptr ? 7 : 3);
}

}  // namespace mli

extern "C" int mli_decode_scan_paged(const float* q_output, const void* const* page_table, const int* lengths,
                                     float* qkt_output, float* attention_result, int n_batch, int n_sequence,
                                     int emb_dim, int elem_bf16, int phases, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    { const mli::WsBody body = mli::ws_body(workspace, workspace_bytes); workspace = body.ptr; workspace_bytes = body.bytes; }
    if (phases < 1 || phases > 7 || phases == 4) return MLI_ERR_BAD_ARG;
    if (!(phases & 4) && qkt_output == nullptr) return MLI_ERR_BAD_ARG;
    hipStream_t st = mli::as_stream(stream);
    const int r = elem_bf16 ? mli::launch_fused_decode<mli::ElemBF16>(q_output, page_table, lengths, qkt_output,
                                                                      attention_result, n_batch, n_sequence, emb_dim,
                                                                      workspace, workspace_bytes, st, phases)
                            : mli::launch_fused_decode<mli::ElemF32>(q_output, page_table, lengths, qkt_output,
                                                                     attention_result, n_batch, n_sequence, emb_dim,
                                                                     workspace, workspace_bytes, st, phases);
    if (r == 1) return 0;
    if (r == 0) return MLI_ERR_BAD_ARG;  // shape not covered by the single-pass kernel (emb_dim too wide) or no workspace
    return r < 0 ? r : r - 1;
}

namespace mli {

}  // namespace mli

#ifdef MLI_SCAN_TRACE
extern "C" int mli_debug_scan_trace(unsigned long long* host, int n_slots) {
    if (n_slots > mli::kTraceSlots) n_slots = mli::kTraceSlots;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mli::mli_scan_trace), (size_t)n_slots * 8 * sizeof(unsigned long long));
}
extern "C" int mli_debug_scan_trace_clear(void) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(mli::mli_scan_trace));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * mli::kTraceSlots * 8);
}
#endif
