// Single-pass decode attention over paged KV, "equal page shares" form -- the lean composition's scan for batches that
// fill the chip.  Same arithmetic per page as attention_fused.hip (a wave owns whole pages, K rows then V rows, online
// softmax, rolling register prefetch through a buffer descriptor); what differs is WHO gets WHICH pages.
//
// attention_fused.hip cuts every row into fixed chunks and launches one workgroup per (row, chunk).  On ragged lengths
// that costs three things (tools/scan_trace.py, config 4): the workgroups of chunks beyond a row's length come and go
// without work (the mean number of working workgroups per CU falls from 1.9 to 1.3 over the launch), every working
// workgroup pays its own prologue and pipeline fill (first page 18 us against 1.8 us per page afterwards, eight times per
// workgroup slot), and the XCDs -- which get the workgroups round-robin, whatever their load -- finish 35 us apart, the
// last 48 us running at 1.5 TB/s.
//
// Here the pages of ALL rows form one sequence (row 0's pages, row 1's, ...), G = 2 x CUs persistent-sized workgroups
// take equal contiguous shares of it, and every workgroup streams its share from its first page to its last with the
// prefetch running across row boundaries.  A share covers the end of one row, possibly some whole rows, and the start of
// another: each such piece ("segment") yields one (max, sum, partial output) triple; a row's triples -- one per
// workgroup that touched it -- are merged in token order by the workgroup whose arrival completes the row (the same
// write-through / arrival-counter / acquire hand-off as attention_fused.hip's lean mode).  Every workgroup finds its
// share by itself: a prefix sum of the rows' page counts in LDS (n_batch <= 2048), no pre-pass launch, no atomics for
// work distribution, the same result for the same lengths whatever the scheduling.
//
// Replaces, inside mli_paged_attention_lean / mli_decode_scan_paged(lean), launch_qkt_paged_attention +
// launch_softmax_in_place_with_lengths + launch_softmax_v_paged_attention (reference paged_attention.cu:270-345).
#include "scan_common.hpp"

namespace mli {

constexpr int kStThreads = 256;
constexpr int kStWaves = kStThreads / kWave;
constexpr int kStMaxRows = 2048;    // rows whose page counts fit the LDS prefix array
constexpr int kStMinPages = 16;     // a workgroup's share is never shorter than this many pages (fewer workgroups work then)
constexpr int kStMinSequence = 1024;
// Triples per row.  A static share is at least s_min = floor(kStMinPages * (100 - dyn_pct) / 100) pages long (G workgroups
// share Ps >= P * (1 - dyn_pct / 100) pages and G <= P / kStMinPages), a granule exactly `gran` (the last one apart).  A row
// of W pages lies in a page interval whose static part meets at most Ls / s_min + 2 shares and whose dynamic part at most
// Ld / gran + 2 granules, Ls + Ld <= W: at most W / min(s_min, gran) + 4 triples.  The workspace has W / 4 slots per row
// (one per 64 tokens); stream_triples_bound() is what launch_stream_decode holds against that before every launch (default
// 4 % / 64 pages: 4 + 4 of 16 at S = 1024; 60 % / 16 pages: 10 + 4).

// -DMLI_SCAN_TRACE: every workgroup records when it passed five points (100 MHz wall clock), read back by
// mli_debug_stream_trace -- a diagnostic build for tools/stream_trace.py, never the product.
#ifdef MLI_SCAN_TRACE
constexpr int kStTraceSlots = 1024;
__device__ unsigned long long mli_stream_trace[kStTraceSlots * 8];
#define MLI_ST_TRACE(i) do { if (threadIdx.x == 0 && blockIdx.x < kStTraceSlots) mli_stream_trace[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MLI_ST_TRACE(i) do { } while (0)
#endif

int nt_loads_for(int B, int S, int D, int esize);             // attention_scan.hip
size_t stats_region_bytes_for(int B, int S);

// MAXSEG = segments (rows touched) a workgroup finishes per group: their wave partials wait in LDS for the group's merge
// RPI = token slots per load instruction (scan_common.hpp; 1 for the reference's fp32 and for bf16), TBR = load
// instructions per batch: a batch covers TBR * RPI slots, a page is 2 * 16 / (TBR * RPI) batches
template <class E, int NJ, bool NT, int TBR, int MAXSEG, int RPI = 1>
__global__ __launch_bounds__(kStThreads, 2) void fused_decode_stream_kernel(
    const float* __restrict__ q, const void* const* __restrict__ page_table, const int* __restrict__ lengths,
    float* __restrict__ out, float2* ml, float* partial, unsigned* arrivals, unsigned* ticket, int B, int S, int D,
    int ml_per_row, int max_pages_wg, int dyn_pct, int gran) {
    constexpr int EPL = E::EPL;
    constexpr int LPR = kWave / RPI;         // lanes per token row
    static_assert(RPI == 1 || NJ == 1, "several rows per instruction only for rows of one lane load");
    constexpr int kRowF = NJ * LPR * EPL;    // floats one wave contributes to a segment's partial
    extern __shared__ __align__(16) unsigned char st_smem[];
    // LDS: prefix[B + 1] | page pointers of the share | q of the group's rows | wave partials | wave (m, l) | lengths of the group's rows
    int* prefix = reinterpret_cast<int*>(st_smem);
    const size_t prefix_bytes = ((size_t)(B + 1) * sizeof(int) + 15) & ~(size_t)15;
    const void** ptr_sh = reinterpret_cast<const void**>(st_smem + prefix_bytes);
    const size_t ptr_bytes = ((size_t)max_pages_wg * 8 + 15) & ~(size_t)15;
    float* q_sh = reinterpret_cast<float*>(st_smem + prefix_bytes + ptr_bytes);              // [MAXSEG][D]
    float* red = q_sh + (size_t)MAXSEG * D;                                                   // [MAXSEG][waves][kRowF]
    float2* wave_ml = reinterpret_cast<float2*>(red + (size_t)MAXSEG * kStWaves * kRowF);     // [MAXSEG][waves]
    int* seg_len = reinterpret_cast<int*>(wave_ml + MAXSEG * kStWaves);                       // [MAXSEG]
    int* misc = seg_len + MAXSEG;   // [8] wave totals of the prefix scan | per published row of a group: row, arrivals, flag
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int W = S / kPage;
    const int w = blockIdx.x, G_launch = gridDim.x;
    MLI_ST_TRACE(0);

    // ---- every workgroup: prefix sums of the rows' page counts (cooperative scan) ----
    {
        const int per = (B + kStThreads - 1) / kStThreads;   // <= kStMaxRows / kStThreads
        int local[kStMaxRows / kStThreads];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < kStMaxRows / kStThreads; ++j) {
            const int b = tid * per + j;
            int pages = 0;
            if (j < per && b < B) pages = (min(max(lengths[b], 0), S) + kPage - 1) / kPage;
            local[j] = sum;
            sum += pages;
        }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int up = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += up;
        }
        if (lane == kWave - 1) misc[wave] = incl;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < kStWaves; ++k)
            if (k < wave) base += misc[k];
        base += incl - sum;
#pragma unroll
        for (int j = 0; j < kStMaxRows / kStThreads; ++j) {
            const int b = tid * per + j;
            if (j < per && b < B) prefix[b] = base + local[j];
        }
        if (tid == kStThreads - 1) prefix[B] = base + sum;
        __syncthreads();
    }
    const long long P = prefix[B];
    // empty rows have no pages, so no workgroup meets them: their zero result is written by a fixed owner
    for (int b = w; b < B; b += G_launch)
        if (prefix[b + 1] == prefix[b])
            for (int i = tid; i < D; i += kStThreads) out[(int64_t)b * D + i] = 0.f;
    if (P == 0) return;
    const long long G = min((long long)G_launch, max(1LL, P / kStMinPages));   // workgroups that get a share
    if (w >= G) return;
    // The page sequence is cut in two.  [0, Ps): equal static shares, one per workgroup -- no coordination at all.
    // [Ps, P): granules of `gran` pages that the workgroups draw from a ticket counter once their share is done.  Equal
    // shares do not finish together: with all 512 workgroups streaming the same number of pages, per-workgroup rates
    // spread by +-15 % (tools/stream_trace.py: the workgroups of every second XCD are ~8 % slower, stream ends 466 ..
    // 645 us), so the fast ones take more of the tail.  dyn_pct = 0 turns the second part off.
    const long long Ps = dyn_pct > 0 && P * dyn_pct / 100 >= (long long)gran ? P - P * dyn_pct / 100 : P;
    const int n_gran = (int)((P - Ps + gran - 1) / gran);
    int lo = (int)((w * Ps) / G), hi = (int)(((w + 1) * Ps) / G);   // this piece's pages; [lo, hi) changes per piece
    int piece = -1;                                                   // -1: the static share, >= 0: granule index
    // row of global page index g: the last b with prefix[b] <= g (rows without pages share a prefix value)
    auto row_of = [&](int g) {
        int a = 0, z = B;
        while (z - a > 1) {
            const int mid = (a + z) >> 1;
            if (prefix[mid] <= g) a = mid;
            else z = mid;
        }
        return a;
    };
    // the triples of row b, in token order: one per static share that meets it, then one per granule that meets it
    auto row_triples = [&](int b, int& w_first, int& n_static, int& k_first, int& n_dyn) {
        const long long R = prefix[b], Eb = prefix[b + 1];
        const long long sR = min(R, Ps), sE = min(Eb, Ps), dR = max(R, Ps), dE = max(Eb, Ps);
        w_first = 0; n_static = 0; k_first = 0; n_dyn = 0;
        if (sE > sR) {   // share k starts at floor(k Ps / G)
            w_first = (int)(((sR + 1) * G + Ps - 1) / Ps) - 1;
            n_static = (int)((sE * G + Ps - 1) / Ps) - 1 - w_first + 1;
        }
        if (dE > dR) {
            k_first = (int)((dR - Ps) / gran);
            n_dyn = (int)((dE - 1 - Ps) / gran) - k_first + 1;
        }
    };
    MLI_ST_TRACE(1);

    const int Du = D / EPL;
    const float scale = sqrtf((float)D);
    const int64_t row_bytes = (int64_t)3 * D * E::kBytes;
    const int64_t seg_bytes = (int64_t)D * E::kBytes;
    const int block_bytes = kPage * 3 * D * E::kBytes;
    bool live[NJ];
    unsigned voff[NJ];
    const int lane_u = lane % LPR, lane_grp = lane / LPR;   // unit inside the row, row inside the load instruction
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int u = lane_u + j * kWave;
        live[j] = u < Du;
        // beyond the row: outside the descriptor's range, reads zeros
        voff[j] = live[j] ? (unsigned)u * 16u + (unsigned)lane_grp * (unsigned)row_bytes : 0x40000000u;
    }
    constexpr int NB = 16 / (TBR * RPI);
    constexpr int NPOS = 2 * NB;
    constexpr int PD = 3;
    fu_u32x4 buf[4][TBR][NJ];
    auto page_ptr = [&](int pi) {
        return reinterpret_cast<const char*>(wave_uniform(reinterpret_cast<const float*>(ptr_sh[pi])));
    };
    auto issue = [&](auto POS, const char* pg) {
        constexpr int pos = decltype(POS)::value;
        constexpr int bi = pos % 4;
        const char* upg = reinterpret_cast<const char*>(wave_uniform(reinterpret_cast<const float*>(pg)));
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(upg), 0, upg != nullptr ? block_bytes : 0, 0x00020000);
        const int base = (pos < NB ? (int)seg_bytes : 2 * (int)seg_bytes) + (pos % NB) * TBR * RPI * (int)row_bytes;
#pragma unroll
        for (int t = 0; t < TBR; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                buf[bi][t][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j], base + t * RPI * (int)row_bytes, NT ? 2 : 0);
    };

    typedef unsigned long long __attribute__((address_space(1)))* gu64_ptr;
    typedef unsigned __attribute__((address_space(1)))* gu32_ptr;

#ifdef MLI_SCAN_TRACE
    int trace_pages = 0, trace_pieces = 0;
#endif
  for (;;) {   // pieces: the static share, then granules while tickets last
   if (hi > lo) {
#ifdef MLI_SCAN_TRACE
    trace_pages += hi - lo;
    ++trace_pieces;
#endif
    const int npages = hi - lo;   // <= max_pages_wg
    const int b_first = row_of(lo), b_last = row_of(hi - 1);
    for (int i = tid; i < npages; i += kStThreads) {
        const int g = lo + i, b = row_of(g);
        ptr_sh[i] = page_table[(int64_t)b * W + (g - prefix[b])];
    }
    // (the barrier of the first group below covers ptr_sh)
    // ---- groups of up to MAXSEG consecutive rows of the piece ----
    for (int r0 = b_first; r0 <= b_last; r0 += MAXSEG) {
        const int r1 = min(r0 + MAXSEG - 1, b_last);
        // this group's pages, as indices into the share: [pg_lo, pg_hi)
        const int pg_lo = max(prefix[r0], lo) - lo, pg_hi = min(prefix[r1 + 1], hi) - lo;
        for (int i = tid; i < (r1 - r0 + 1) * D; i += kStThreads) q_sh[i] = q[(int64_t)r0 * D + i];   // rows r0 .. r1 are adjacent in q
        if (tid <= r1 - r0) seg_len[tid] = min(max(lengths[r0 + tid], 0), S);
        if (tid < MAXSEG * kStWaves) wave_ml[tid] = make_float2(-INFINITY, 0.f);
        __syncthreads();

        // ---- the wave streams its pages of the group: pg_lo + wave, + 4, ... ----
        float qr[NJ][EPL];
        float run_m = -INFINITY, run_l = 0.f;
        float acc[NJ][EPL];
        int cur = -1;          // row the wave's state belongs to (-1: none yet)
        int cur_end = pg_lo;   // first share-local page index beyond that row
        auto flush = [&]() {
            if (cur < 0) return;
            const int s = cur - r0;
            if (lane == 0) wave_ml[s * kStWaves + wave] = make_float2(run_m, run_l);
            float* dst = red + ((size_t)s * kStWaves + wave) * kRowF;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float a = rpi_group_sum<RPI>(acc[j][e]);   // the lane groups hold different slots' contributions
                    if (RPI == 1 || lane < LPR) dst[(j * kWave + lane_u) * EPL + e] = a;
                }
        };
        const int p_first = pg_lo + wave;
        const char* page = p_first < pg_hi ? page_ptr(p_first) : nullptr;
        if (p_first < pg_hi) {
            issue(std::integral_constant<int, 0>{}, page);
            issue(std::integral_constant<int, 1>{}, page);
            issue(std::integral_constant<int, 2>{}, page);
        }
        for (int pi = p_first; pi < pg_hi; pi += kStWaves) {
            const bool has_next = pi + kStWaves < pg_hi;
            const char* next = has_next ? page_ptr(pi + kStWaves) : nullptr;
            if (pi >= cur_end) {   // the wave enters another row (wave-uniform): park its state, take the new row's q
                flush();
                int b = cur < 0 ? r0 : cur + 1;
                while (min(prefix[b + 1], hi) - lo <= pi) ++b;   // rows without pages (or without pages for this wave) are skipped
                cur = b;
                cur_end = min(prefix[b + 1], hi) - lo;
                run_m = -INFINITY;
                run_l = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        acc[j][e] = 0.f;
                        qr[j][e] = live[j] ? q_sh[(size_t)(b - r0) * D + (lane_u + j * kWave) * EPL + e] : 0.f;
                    }
            }
            // live tokens of this page: it is page (lo + pi - prefix[cur]) of row cur
            const int nt = min(kPage, seg_len[cur - r0] - (lo + pi - prefix[cur]) * kPage);
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
                    if (has_next) issue(std::integral_constant<int, tgt - NPOS>{}, next);
                }
                if constexpr (pos < NB) {
#pragma unroll
                    for (int t = 0; t < TBR; ++t)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) ElemMath<E>::dot(buf[bi][t][j], qr[j], sacc[pos * TBR + t]);
                    if constexpr (pos == NB - 1) {
                        const float tot = rpi_reduce<RPI>(sacc, lane);  // lane holds the sum for slot (lane >> 2) & 15 (RPI = 1)
                        const int slot = rpi_slot_of_lane<RPI>(lane);
                        const bool valid = slot < nt;
                        const float score = tot / scale;
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
                    constexpr int first = (pos - NB) * TBR;
#pragma unroll
                    for (int t = 0; t < TBR; ++t) {
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
            if (pi == p_first && r0 == b_first && piece < 0) MLI_ST_TRACE(2);
#endif
        }
        flush();
        if (piece < 0) MLI_ST_TRACE(3);
        __syncthreads();

        // ---- one triple per row of the group: merge the waves (fixed order); a row that lies wholly inside this share
        //      gets its final result, the others publish their triple.  The publications of a group share ONE drain,
        //      barrier and round of arrival counts (all workgroups reach this point at about the same time -- their
        //      shares are equal -- so nothing else would hide a per-row hand-off latency here) ----
        int n_pub = 0;   // (workgroup-uniform)
        for (int b = r0; b <= r1; ++b) {
            const int R = prefix[b], Pb = prefix[b + 1] - R;
            if (Pb == 0) continue;
            const int s = b - r0;
            int w_first, n_static, k_first, n_dyn;
            row_triples(b, w_first, n_static, k_first, n_dyn);
            const int nseg = n_static + n_dyn, slot = piece < 0 ? w - w_first : n_static + piece - k_first;
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < kStWaves; ++k) m = fmaxf(m, wave_ml[s * kStWaves + k].x);
            float wsc[kStWaves];
            float l = 0.f;
#pragma unroll
            for (int k = 0; k < kStWaves; ++k) {
                const float2 v = wave_ml[s * kStWaves + k];
                wsc[k] = v.x == -INFINITY ? 0.f : expf(v.x - m);
                l += v.x == -INFINITY ? 0.f : v.y * wsc[k];
            }
            const bool whole = nseg == 1;
            float* o = whole ? out + (int64_t)b * D : partial + ((int64_t)b * ml_per_row + slot) * D;
            const float norm = whole ? 1.f / l : 1.f;
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(o, 0, D * (int)sizeof(float), 0x00020000);
            const float* rs = red + (size_t)s * kStWaves * kRowF;
            for (int i = 4 * tid; i < D; i += 4 * kStThreads) {
                float r[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < kStWaves; ++k)
                        if (wsc[k] != 0.f) t += rs[k * kRowF + i + k4] * wsc[k];   // a wave without pages here left no data
                    r[k4] = t * norm;
                }
                if (whole) {
                    *reinterpret_cast<float4*>(o + i) = make_float4(r[0], r[1], r[2], r[3]);
                } else {
                    fu_u32x4 raw;
                    raw.x = __float_as_uint(r[0]); raw.y = __float_as_uint(r[1]); raw.z = __float_as_uint(r[2]); raw.w = __float_as_uint(r[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(raw, orow, i * (int)sizeof(float), 0, 16);   // write-through (sc1)
                }
            }
            if (whole) continue;
            if (tid == 0) {
                const unsigned long long packed = ((unsigned long long)__float_as_uint(l) << 32) | __float_as_uint(m);
                __hip_atomic_store((gu64_ptr)(ml + (int64_t)b * ml_per_row + slot), packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                misc[8 + n_pub] = b;          // rows published by this group ...
                misc[8 + MAXSEG + n_pub] = nseg;  // ... and the arrivals each of them waits for
            }
            ++n_pub;
        }
        if (n_pub > 0) {
            // (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload, every storing wave's vmcnt(0), barrier, counter
            // add; consumer: agent acquire, vmcnt(0), barrier, then loads)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < n_pub) {
                const int b = misc[8 + tid], nseg = misc[8 + MAXSEG + tid];
                const unsigned before = __hip_atomic_fetch_add((gu32_ptr)(arrivals + b), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = before + 1u == (unsigned)nseg;
                if (last) {
                    __hip_atomic_store((gu32_ptr)(arrivals + b), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                misc[8 + 2 * MAXSEG + tid] = last;
            }
            __syncthreads();
            for (int k = 0; k < n_pub; ++k) {
                if (!misc[8 + 2 * MAXSEG + k]) continue;   // (workgroup-uniform)
                // this workgroup completed row b: its triples in slot (= token) order; statistics through LDS (the q buffer
                // of this group is done with)
                const int b = misc[8 + k], nseg = misc[8 + MAXSEG + k];
                float2* ml_row = ml + (int64_t)b * ml_per_row;
                float2* ml_sh = reinterpret_cast<float2*>(q_sh);   // nseg <= ml_per_row <= MAXSEG * D / 2 (checked by the launcher)
                for (int i = tid; i < nseg; i += kStThreads) {
                    const unsigned long long packed = __hip_atomic_load((gu64_ptr)(ml_row + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ml_sh[i] = make_float2(__uint_as_float((unsigned)packed), __uint_as_float((unsigned)(packed >> 32)));
                }
                __syncthreads();
                float mm = -INFINITY;
                for (int i = 0; i < nseg; ++i) mm = fmaxf(mm, ml_sh[i].x);
                float ll = 0.f;
                for (int i = 0; i < nseg; ++i) ll = fmaf(ml_sh[i].y, expf(ml_sh[i].x - mm), ll);
                const float inv_l = 1.f / ll;
                const float* pr = partial + (int64_t)b * ml_per_row * D;
                for (int d = 4 * tid; d < D; d += 4 * kStThreads) {
                    float r[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int i0 = 0; i0 < nseg; i0 += 8) {
                        fu_u32x4 v[8];
#pragma unroll
                        for (int k8 = 0; k8 < 8; ++k8) {
                            if (i0 + k8 < nseg) {
                                const float* row_i = pr + (int64_t)(i0 + k8) * D;
                                const __amdgpu_buffer_rsrc_t prow =
                                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row_i), 0, D * (int)sizeof(float), 0x00020000);
                                v[k8] = __builtin_amdgcn_raw_buffer_load_b128(prow, d * (int)sizeof(float), 0, 16);
                            }
                        }
#pragma unroll
                        for (int k8 = 0; k8 < 8; ++k8) {
                            if (i0 + k8 < nseg) {
                                const float wgt = expf(ml_sh[i0 + k8].x - mm);
                                r[0] = fmaf(__uint_as_float(v[k8].x), wgt, r[0]);
                                r[1] = fmaf(__uint_as_float(v[k8].y), wgt, r[1]);
                                r[2] = fmaf(__uint_as_float(v[k8].z), wgt, r[2]);
                                r[3] = fmaf(__uint_as_float(v[k8].w), wgt, r[3]);
                            }
                        }
                    }
                    *reinterpret_cast<float4*>(out + (int64_t)b * D + d) = make_float4(r[0] * inv_l, r[1] * inv_l, r[2] * inv_l, r[3] * inv_l);
                }
                __syncthreads();   // ml_sh is reused by the next completed row
            }
        }
        __syncthreads();   // q_sh, the wave partials and misc are reused by the next group
    }
   }
    // ---- next piece: a granule of the dynamic part, if any is left ----
    if (n_gran == 0) break;
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add((gu32_ptr)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every sharing workgroup draws until it gets a ticket beyond the granules: n_gran + G draws in all, the last
        // one puts the counter back to zero for the next launch
        if (t + 1u == (unsigned)(n_gran + G)) __hip_atomic_store((gu32_ptr)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        misc[8 + 3 * MAXSEG] = (int)t;
    }
    __syncthreads();
    piece = misc[8 + 3 * MAXSEG];
    __syncthreads();
    if (piece >= n_gran) break;
    lo = (int)(Ps + (long long)piece * gran);
    hi = (int)min((long long)lo + gran, P);
  }
    MLI_ST_TRACE(4);
#ifdef MLI_SCAN_TRACE
    if (threadIdx.x == 0 && blockIdx.x < kStTraceSlots) {
        mli_stream_trace[blockIdx.x * 8 + 5] = (unsigned long long)trace_pages;
        mli_stream_trace[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // XCC_ID
        mli_stream_trace[blockIdx.x * 8 + 7] = (unsigned long long)trace_pieces;
    }
#endif
}

static thread_local int g_scan_stream = 1;   // mli_tune "scan_stream": 1 (default) = lean scans of chip-filling batches take the equal-share form
void set_scan_stream(int v) { g_scan_stream = v != 0; }

// mli_tune "scan_stream_dynamic_pct" / "scan_stream_granule": the share of the page sequence (per cent, 0 = none) that is
// handed out dynamically in granules of that many pages after the equal static shares
// Measured at config 4 (one box, lean scan, us): bf16 chunked grid 690.8 | shares only 671.1 | 4 % in 64-page granules
// 662.4 | 12 % in 16-page granules 676.9; fp32 1276.9 | 1265.2 | 1233.6 | 1247.7.  The gain of the dynamic part is small
// because the memory system, not the slowest workgroup, sets the pace: 512 workgroups keep 49 MB in flight, several times
// what saturates HBM, so the workgroups that are still streaming simply speed up when others finish.
static thread_local int g_stream_dyn_pct = 4;
static thread_local int g_stream_granule = 64;
void set_stream_dyn_pct(int v) { g_stream_dyn_pct = v < 0 ? 0 : (v > 60 ? 60 : v); }
void set_stream_granule(int v) { g_stream_granule = v < 16 ? 16 : (v > 256 ? 256 : v); }
static thread_local int g_scan_stream_min = 1 << 21;   // mli_tune "scan_stream_min_tokens": n_batch * n_sequence from which it is used
void set_scan_stream_min(int v) { g_scan_stream_min = v < 0 ? 0 : v; }

// worst-case number of triples a row can have under the current split (see the header comment)
static int stream_triples_bound(int W) {
    const int s_min = std::max(1, kStMinPages * (100 - g_stream_dyn_pct) / 100);
    const int piece = g_stream_dyn_pct > 0 ? std::min(s_min, g_stream_granule) : s_min;
    return W / piece + 4;
}

template <class E>
bool stream_decode_applies(int B, int S, int D) {
    const int Du = D / E::EPL;
    const int nj = ceil_div_i(Du, kWave);
    if (!g_scan_stream || nj > 2 || D % E::EPL != 0 || S % kPage != 0 || S < kStMinSequence || B > kStMaxRows) return false;
    if (stream_triples_bound(S / kPage) > ceil_div_i(S, 64)) return false;   // a row's triples would not fit its workspace slots
    // worth it only where the batch fills the chip: below ~2 full rounds of 512-token chunks the chunked grid is as good
    return (int64_t)B * S >= g_scan_stream_min;
}
template bool stream_decode_applies<ElemF32>(int, int, int);
template bool stream_decode_applies<ElemBF16>(int, int, int);
template bool stream_decode_applies<ElemFP8>(int, int, int);

// 1 = ran, 0 = not applicable (the caller takes the chunked kernel), else an error (+1 if positive)
template <class E>
int launch_stream_decode(const float* q, const void* const* page_table, const int* lengths, float* out, int B, int S, int D,
                         void* ws, size_t ws_bytes, hipStream_t st) {
    const int Du = D / E::EPL;
    const int nj = ceil_div_i(Du, kWave);
    if (!stream_decode_applies<E>(B, S, D)) return 0;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // Two workgroups per CU.  (Tried for the fp8 kernel at emb_dim 512, whose batches are half as many bytes as the bf16
    // kernel's: three / four workgroups per CU under __launch_bounds__(256, 3 / 4) -- 168 / 128 VGPRs, 57 / 134 of them
    // spilled -- 543 / 894 us against 338; decoding and multiplying in pairs (v_pk_fma_f32): 338.8 against 338.0 us.)
    const int G = 2 * n_cu;
    const int ml_per_row = ceil_div_i(S, 64);
    const size_t stats_bytes = stats_region_bytes_for(B, S);
    if (ws == nullptr || ws_bytes < stats_bytes + (size_t)B * ml_per_row * D * sizeof(float)) return 0;
    float2* ml = reinterpret_cast<float2*>(ws);
    float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + stats_bytes);
    unsigned* arrivals = ws_arrivals(ws);
    // a share is at most ceil(P / G) pages where every workgroup works, and below 2 * kStMinPages otherwise
    const int64_t p_max = (int64_t)B * (S / kPage);
    const int max_pages_wg = (int)std::max<int64_t>(std::max<int64_t>((p_max + G - 1) / G + 1, 2 * kStMinPages + 1), g_stream_granule);
    // fp8 rows narrower than one load instruction: 2 or 4 token slots per instruction (scan_common.hpp)
    const int rpi = std::is_same<E, ElemFP8>::value ? (Du <= 16 ? 4 : Du <= 32 ? 2 : 1) : 1;
    const int kRowF = nj * (kWave / rpi) * E::EPL;
    const int maxseg = kRowF <= 512 ? 4 : 2;
    const size_t smem = (((size_t)(B + 1) * sizeof(int) + 15) & ~(size_t)15) + (((size_t)max_pages_wg * 8 + 15) & ~(size_t)15) +
                        (size_t)maxseg * D * sizeof(float) + (size_t)maxseg * kStWaves * kRowF * sizeof(float) +
                        (size_t)maxseg * kStWaves * sizeof(float2) + (size_t)maxseg * sizeof(int) + (8 + 3 * (size_t)maxseg + 1) * sizeof(int);
    if ((size_t)ml_per_row * sizeof(float2) > (size_t)maxseg * D * sizeof(float) || smem > 80 * 1024) return 0;
    const bool nt = nt_loads_for(B, S, D, E::kBytes);
#define MLI_ST_LAUNCH(NJ, NT, TBR, MAXSEG, ...)                                                                         \
    do {                                                                                                                 \
        auto kern = fused_decode_stream_kernel<E, NJ, NT, TBR, MAXSEG, ##__VA_ARGS__>;                                   \
        if (smem > 64 * 1024) {                                                                                          \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess) return (int)e + 1;                                                                      \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, dim3(G), dim3(kStThreads), smem, st, q, page_table, lengths, out, ml, partial, arrivals, \
                           arrivals + (kMaxArrivalRows - 1), B, S, D, ml_per_row, max_pages_wg, g_stream_dyn_pct,     \
                           g_stream_granule);                                                                            \
    } while (0)
    if constexpr (std::is_same<E, ElemFP8>::value) {
        if (rpi == 4) {
            if (nt) MLI_ST_LAUNCH(1, true, 2, 4, 4);
            else MLI_ST_LAUNCH(1, false, 2, 4, 4);
        } else if (rpi == 2) {
            if (nt) MLI_ST_LAUNCH(1, true, 4, 4, 2);
            else MLI_ST_LAUNCH(1, false, 4, 4, 2);
        } else if (nj == 1) {
            if (nt) MLI_ST_LAUNCH(1, true, 8, 2);
            else MLI_ST_LAUNCH(1, false, 8, 2);
        } else {
            if (nt) MLI_ST_LAUNCH(2, true, 4, 2);
            else MLI_ST_LAUNCH(2, false, 4, 2);
        }
    } else if (nj == 1) {
        if (nt) MLI_ST_LAUNCH(1, true, 8, 4);
        else MLI_ST_LAUNCH(1, false, 8, 4);
    } else if (kRowF <= 512) {
        if (nt) MLI_ST_LAUNCH(2, true, 4, 4);
        else MLI_ST_LAUNCH(2, false, 4, 4);
    } else {
        if (nt) MLI_ST_LAUNCH(2, true, 4, 2);
        else MLI_ST_LAUNCH(2, false, 4, 2);
    }
#undef MLI_ST_LAUNCH
    const int rc = launch_status();
    return rc ? (rc > 0 ? rc + 1 : rc) : 1;
}

template int launch_stream_decode<ElemF32>(const float*, const void* const*, const int*, float*, int, int, int, void*, size_t, hipStream_t);
template int launch_stream_decode<ElemBF16>(const float*, const void* const*, const int*, float*, int, int, int, void*, size_t, hipStream_t);
template int launch_stream_decode<ElemFP8>(const float*, const void* const*, const int*, float*, int, int, int, void*, size_t, hipStream_t);

}  // namespace mli

#ifdef MLI_SCAN_TRACE
extern "C" int mli_debug_stream_trace(unsigned long long* host, int n_slots) {
    if (n_slots > mli::kStTraceSlots) n_slots = mli::kStTraceSlots;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mli::mli_stream_trace), (size_t)n_slots * 8 * sizeof(unsigned long long));
}
#endif
