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
#include "scan_item_body.hpp"

namespace mli {

int sv_chunk_tokens_for(int n_batch, int n_sequence);  // attention_scan.hip
int tuned_chunk_tokens();
template <class E>
int launch_stream_decode(const float* q, const void* const* page_table, const int* lengths, float* out, int B, int S, int D,
                         void* ws, size_t ws_bytes, hipStream_t st);   // attention_stream.hip
template <class E>
bool stream_decode_applies(int B, int S, int D);
size_t stats_region_bytes_for(int B, int S);
int nt_loads_for(int B, int S, int D, int esize);

// One workgroup per row (short sequences: the whole row is one item): workgroups start in grid order and the rows' lengths
// are ragged, so whichever long rows happen to sit at the end of the grid run alone at the end (README workload, B=1024,
// S=128, D=2048: 5.3 TB/s against 6.4 with equal lengths).  This hands the rows out LONGEST FIRST instead: workgroup r takes
// the row of rank r by page count (descending; equal counts in row order).  Every workgroup derives the same ranking from
// the lengths -- a histogram over the page counts, then the j-th row of its bucket by a block-wide count --: ~2 us of
// prologue per workgroup, no pre-pass, deterministic.  All kFuThreads threads call it; n_batch <= kMaxOrderedRows.
constexpr int kMaxOrderedRows = 2048;
constexpr int kMaxOrderedPages = 64;
__device__ __forceinline__ int longest_first_row(const int* __restrict__ lengths, int n_batch, int S, int rank) {
    __shared__ int hist[kMaxOrderedPages + 1];
    __shared__ int wave_cnt[kFuWaves];
    __shared__ int found_row;
    constexpr int kPer = kMaxOrderedRows / kFuThreads;
    const int tid = threadIdx.x;
    const int per = (n_batch + kFuThreads - 1) / kFuThreads;  // rows per thread, a contiguous segment
    if (tid <= kMaxOrderedPages) hist[tid] = 0;
    __syncthreads();
    int pages[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int row = tid * per + j;
        pages[j] = -1;
        if (j < per && row < n_batch) {
            pages[j] = (min(max(lengths[row], 0), S) + kPage - 1) / kPage;
            atomicAdd(&hist[pages[j]], 1);
        }
    }
    __syncthreads();
    // the bucket of this rank (page counts descending) and the rank inside it
    int bucket = 0, before = 0;
    for (int p = kMaxOrderedPages; p >= 0; --p) {
        const int h = hist[p];
        if (rank < before + h) {
            bucket = p;
            break;
        }
        before += h;
    }
    const int j_in_bucket = rank - before;
    // the j-th row of the bucket in row order: matches per thread segment, exclusive prefix over the threads
    int mine = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) mine += pages[j] == bucket;
    int incl = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off, kWave);
        if ((tid & (kWave - 1)) >= off) incl += up;
    }
    if ((tid & (kWave - 1)) == kWave - 1) wave_cnt[tid / kWave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < tid / kWave; ++w) base += wave_cnt[w];
    if (j_in_bucket >= base && j_in_bucket < base + mine) {
        int seen = base;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            if (pages[j] == bucket) {
                if (seen == j_in_bucket) found_row = tid * per + j;
                ++seen;
            }
        }
    }
    __syncthreads();
    return found_row;
}

template <class E, int NJ, bool NT, int TBR, int MINW, int WAVES, bool DS = false, bool SCORES = true, int RPI = 1>
__global__ __launch_bounds__(WAVES * kWave, MINW) void fused_decode_scan_kernel(
    const float* __restrict__ q, const void* const* __restrict__ page_table, const int* __restrict__ lengths,
    float* __restrict__ qkt, float* __restrict__ out, float2* ml, float* partial,
    int S, int D, int ct, int ml_per_row, int nchunk_max, int direct, unsigned* __restrict__ ticket, int tail,
    int slots, unsigned* arrivals) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
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
    if constexpr (WAVES * kWave == kFuThreads) {
        if (direct == 2) b = longest_first_row(lengths, (int)gridDim.x, S, (int)blockIdx.x);
    }
    if (ticket != nullptr) {
        __shared__ unsigned item_sh;
        if (threadIdx.x == 0) item_sh = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned item = item_sh;
        if (item >= gridDim.x * gridDim.y) return;  // cannot happen with a zeroed counter; never index out of range
        b = (int)(item % gridDim.x);
        c = (int)(item / gridDim.x);
    }
    fused_scan_item<E, NJ, NT, TBR, WAVES, DS, SCORES, RPI>(q, page_table, lengths, qkt, out, ml, partial, S, D, ct, ml_per_row,
                                                             nchunk_max, direct, tail, slots, arrivals, b, c, c == 0,
                                                             (int)gridDim.x, smem_raw);
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

static thread_local int g_flash = 1;
static thread_local int g_flash_variant = 0;  // register-budget variants of the scan kernel (tuning)
// mli_tune "scan_partial_last": 1 (default) = full chunks first, every row's remainder behind them in pieces of
// "scan_tail_tokens" tokens; 0 = plain chunk order
static thread_local int g_partial_last = 1;
void set_partial_last(int v) { g_partial_last = v != 0; }
// mli_tune "scan_tail_tokens": 0 (default) = the whole remainder as one piece, else a power of two in [64, chunk].
// Measured at config 4 (bf16, 512-token chunks, lean form): one piece 658.7 us, 256-token pieces 661.9, 128: 668.7,
// 64: 688.0 -- every item costs about 2.5 us of a workgroup slot (prologue chain lengths -> page pointers -> first K
// rows, epilogue merge and publication), more than the shorter end of the launch gives back.
static thread_local int g_tail_tokens = 0;
void set_tail_tokens(int v) { g_tail_tokens = v; }
// mli_tune "scan_dynamic_items": ticketed (row, chunk) assignment.  Off by default: it shortens the kernel by 0.6-1.5 %
// (4-10 us at config 4), and the hipMemsetAsync that zeroes the counter before every launch costs the stream ~8 us.
static thread_local int g_dynamic_items = 0;
void set_dynamic_items(int v) { g_dynamic_items = v != 0; }
// mli_tune "scan_merge" (lean mode only): 1 (default) = the workgroup that completes a row merges its chunks inside
// the scan launch, 0 = the separate combine launch (bit-identical results)
static thread_local int g_row_order = 1;  // mli_tune "scan_row_order": 0 = one-workgroup-per-row grids take the rows in grid order
void set_row_order(int v) { g_row_order = v != 0; }
static thread_local int g_scan_merge = 1;
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
    constexpr bool kFp8 = std::is_same<E, ElemFP8>::value;
    if (kFp8 && (!lean || nj > 2)) return 0;   // the fp8 extension: lean form, rows of up to two lane loads (emb_dim <= 2048)
    // fp8 rows narrower than one load instruction: 2 or 4 token slots per instruction (scan_common.hpp)
    const int rpi = kFp8 ? (Du <= 16 ? 4 : Du <= 32 ? 2 : 1) : 1;
    const bool dsplit = nj > 2;  // wide rows: the four waves split the row instead of the pages
    const int nj_ds = ceil_div_i(Du, kWave * kFuWaves);  // 1 or 2
    // variant 3: single-wave workgroups of 128 tokens -- every wave is its own scheduling unit, no LDS merge,
    // no barrier; the hardware dispatcher does the load balancing
    const bool solo = g_flash_variant == 3 && S > 128 && !dsplit && !(phases & 4);
    // short sequences with a full batch: one workgroup per row (no partials, no combine launch) beats two 64-token
    // chunks (README workload, S = 128: 200 vs 209 us)
    const int ct = solo ? 128 : (S <= 128 && B >= 256 && tuned_chunk_tokens() == 0) ? 128 : fused_chunk_tokens(B, S);
    const int nchunk = ceil_div_i(S, ct);
    // one workgroup per row: hand the rows out longest first where the batch has more rows than the chip has workgroup slots
    const bool ordered = g_row_order && nchunk == 1 && !solo && B > 512 && B <= kMaxOrderedRows && S / kPage <= kMaxOrderedPages;
    const int direct = nchunk == 1 ? (ordered ? 2 : 1) : 0;
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
    const size_t red_bytes = (dsplit ? (size_t)2 * kFuWaves * 16 : (size_t)waves * nj * (kWave / rpi) * E::EPL) * sizeof(float);
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
    if constexpr (kFp8) {
        if (phases & 1) {
            if (rpi == 4) {
                if (nt) MLI_FU_LAUNCH(1, true, 2, 2, 4, false, false, 4);
                else MLI_FU_LAUNCH(1, false, 2, 2, 4, false, false, 4);
            } else if (rpi == 2) {
                if (nt) MLI_FU_LAUNCH(1, true, 4, 2, 4, false, false, 2);
                else MLI_FU_LAUNCH(1, false, 4, 2, 4, false, false, 2);
            } else if (nj == 1) {
                if (nt) MLI_FU_LAUNCH(1, true, 8, 2, 4, false, false);
                else MLI_FU_LAUNCH(1, false, 8, 2, 4, false, false);
            } else {
                if (nt) MLI_FU_LAUNCH(2, true, 4, 2, 4, false, false);
                else MLI_FU_LAUNCH(2, false, 4, 2, 4, false, false);
            }
        }
    } else if ((phases & 1) && lean) {
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

// fp8 (OCP e4m3) pages: the lean form only
int launch_fused_decode_fp8(const float* q, const uint8_t* const* page_table, const int* lengths, float* out, int B, int S,
                            int D, void* ws, size_t ws_bytes, hipStream_t st) {
    return launch_fused_decode<ElemFP8>(q, reinterpret_cast<const void* const*>(page_table), lengths, nullptr, out, B, S, D, ws,
                                        ws_bytes, st, 7);
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
    if (elem_bf16 < MLI_ELEM_F32 || elem_bf16 > MLI_ELEM_FP8) return MLI_ERR_BAD_ARG;
    if (elem_bf16 == MLI_ELEM_FP8 && !(phases & 4)) return MLI_ERR_BAD_ARG;   // the fp8 extension has the lean form only
    const int r = elem_bf16 == MLI_ELEM_FP8
                      ? mli::launch_fused_decode<mli::ElemFP8>(q_output, page_table, lengths, qkt_output, attention_result,
                                                               n_batch, n_sequence, emb_dim, workspace, workspace_bytes, st, phases)
                  : elem_bf16 ? mli::launch_fused_decode<mli::ElemBF16>(q_output, page_table, lengths, qkt_output,
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
