// One decode step of a SMALL paged batch (BASELINE config 3: B=256, D=256, S=1024, fp32) as ONE launch.
//
// As separate launches the step is  projection (panel GEMM, 6-7 us) | scan (47-50 us) | logits + argmax (7 us) |
// finalize (4-5 us)  plus three kernel boundaries: a third of it is small dependent kernels around a scan that is itself
// short enough for its cold start (first loads of 2048 waves queueing up, 7-8 us) and its drain (the last 8 us run at a
// fraction of the bandwidth) to matter.  Here the four kernels' workgroup bodies are ROLES of one grid:
//
//   ticket order      role                         waits for                              tells
//   0 .. nP-1         projection tile (32 x 32)    --                                     proj[row block] += 1
//   nP .. nP+nS-1     scan item (row, chunk)       proj[block] == tiles of a block        merged[block] += 1 per finished row
//   the rest          logits tile + argmax         merged[block] == rows of the block     head[block] += 1
//                     last logits tile of a block: picks the tokens, writes lengths / next embedding / decoder_result
//
// * A workgroup takes its role from a TICKET (one atomic add at its start), not from its grid position: every role only
//   waits for roles with smaller tickets, and a ticket is only ever held by a workgroup that is already running, so a
//   waiting workgroup never waits for one that has not been dispatched -- whatever order the hardware starts them in.
// * Scan items request the K / V rows of every page except the row's LAST one before they wait (the projection writes
//   the K and V row of token L - 1 only): the scan's cold start overlaps the projection.  Logits tiles request their
//   emb_table panel before they wait and start as soon as THEIR 32 rows are merged: the head overlaps the scan's drain.
// * Hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility): producers store write-through (sc1), every storing
//   wave drains (s_waitcnt vmcnt(0)), barrier, one lane adds to the counter at agent scope; consumers: one lane polls
//   (relaxed sc1 loads, s_sleep), barrier, then loads.  No agent acquire (2-6 us per workgroup under load) where the
//   handed-off bytes are FIRST TOUCHED after the hand-off -- sf_wait's comment says why that is enough; the rows' chunk
//   merges (row_publish_merge), whose partial rows ARE re-read from launch to launch, keep theirs.
// * MEASURED SLOWER than the separate launches (config 3: 72 vs 67 us) and therefore opt-in
//   (mli_tune "step_fused"): the launches were already back to back, and every role is a chain of dependent memory round
//   trips that a hand-off does not shorten.
// * lengths[b] is rewritten by the block's finalizer while scan items of OTHER tickets may not have read it yet (an item
//   beyond a row's length is empty, but it has to read the length to know): every scan item counts in len_read[block]
//   once the length is in its register, and the finalizer waits for that count first.
// * All counters are back at zero when the launch ends (the last finalizer resets them): launches, graph replays and
//   other shapes sharing the workspace need no memset.  Every spin is bounded: on overrun the workgroup records a code
//   in the error word (mli_debug_step_fused_error) and goes on, so the grid always drains.
//
// Same tile bodies, same arithmetic, same merge order as the separate launches: attention_result, tokens, lengths and
// pages are bit-identical to mli_paged_attention_lean + mli_paged_decoder_fused (tests/test_step_fused_gpu.py).
// Replaces, for such batches, the launch sequence of PagedAttentionInferenceModel::forward's decode rounds
// (reference src/inference_model.cpp:56-81: paged_attention + paged_attention_decoder per round).
#include <atomic>
#include <cfloat>

#include "gemm_panel_body.hpp"
#include "scan_item_body.hpp"

namespace mli {

bool plan_chunked_scan_f32(int B, int S, int D, size_t ws_bytes, ChunkedScanPlan* p);  // attention_fused.hip
bool gemm_panel_wanted(int M, int N_total, int K, bool vec4);                          // proj_gemm_panel.hip

// counters, as unsigned indices into the workspace's arrival region; [0, 8192) are the scan's per-row arrival counters.
// Every counter has a 128-byte line of its own: the ticket word takes one returning atomic per workgroup, the others are
// polled -- on a shared line the polls would queue up in front of the atomics.
constexpr int kSfLine = 32;                           // unsigneds per line
constexpr int kSfBase = 8192;
constexpr int kSfTicket = kSfBase + 0 * kSfLine;
constexpr int kSfBlocksDone = kSfBase + 1 * kSfLine;
constexpr int kSfError = kSfBase + 2 * kSfLine;
constexpr int kSfMaxBlocks = 60;                      // row blocks of 32: n_batch <= 1920 (a small-batch path)
constexpr int kSfProj = kSfBase + 4 * kSfLine;                    // [block] projection tiles stored
constexpr int kSfLenRead = kSfProj + kSfMaxBlocks * kSfLine;      // [block] scan items that hold their row's length
constexpr int kSfMerged = kSfLenRead + kSfMaxBlocks * kSfLine;    // [block] rows whose attention_result is written
constexpr int kSfHead = kSfMerged + kSfMaxBlocks * kSfLine;       // [block] logits tiles stored
static_assert(kSfHead + kSfMaxBlocks * kSfLine < kMaxArrivalRows - 1, "counters must fit the arrival region");
constexpr unsigned kSfSpinCap = 1u << 22;             // polls (>= 0.3 us each): seconds, against a step of < 100 us

typedef unsigned __attribute__((address_space(1)))* sf_u32_ptr;

// -DMLI_STEP_TRACE: per-ticket time stamps (100 MHz wall clock) for tools/step_fused_trace.py -- never the product
#ifdef MLI_STEP_TRACE
constexpr int kStepTraceSlots = 8192;
__device__ unsigned long long mli_step_trace[kStepTraceSlots * 8];
#define MLI_STAMP(t, i) do { if (threadIdx.x == 0 && (t) < (unsigned)kStepTraceSlots) mli_step_trace[(t) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MLI_NOTE(t, i, v) do { if (threadIdx.x == 0 && (t) < (unsigned)kStepTraceSlots) mli_step_trace[(t) * 8 + (i)] = (unsigned long long)(v); } while (0)
#else
#define MLI_STAMP(t, i) do { } while (0)
#define MLI_NOTE(t, i, v) do { } while (0)
#endif

__device__ __forceinline__ void sf_add(unsigned* c) {
    (void)__hip_atomic_fetch_add((sf_u32_ptr)c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sf_set(unsigned* c, unsigned v) {
    __hip_atomic_store((sf_u32_ptr)c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: wait until *c >= target (counters only grow until the launch's last workgroup resets them); `code` goes to
// the error word when the wait gives up.  ACQUIRE: then drop this CU's cached copies of the producers' lines (agent
// acquire, 2-6 us under load) -- needed where the consumer may hold such copies.  The scan items do not: the bytes they
// wait for (q_output[b], the K / V row of token L - 1) are written once per launch, write-through, and nothing on any CU
// reads those lines earlier in the launch (the pages requested before the wait are OTHER pages), so no cache can hold
// an older copy; their first read after the matched poll comes from memory.
template <bool ACQUIRE>
__device__ __forceinline__ void sf_wait(unsigned* ctr, int index, unsigned target, unsigned code) {
    unsigned it = 0;
    while (__hip_atomic_load((sf_u32_ptr)(ctr + index), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++it >= kSfSpinCap) {
            sf_set(ctr + kSfError, code);
            break;
        }
        __builtin_amdgcn_s_sleep(ACQUIRE ? 16 : 4);
    }
    if (ACQUIRE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

struct StepScanGate {
    static constexpr bool kGated = true;
    unsigned* ctr;
    unsigned proj_tiles;  // projection tiles of one row block
    unsigned ticket;
    __device__ __forceinline__ void length_read(int b) const { sf_add(ctr + kSfLenRead + (b / PM) * kSfLine); }
    __device__ __forceinline__ void wait_inputs(int b) const {
        MLI_STAMP(ticket, 2);
        if (threadIdx.x == 0) sf_wait<false>(ctr, kSfProj + (b / PM) * kSfLine, proj_tiles, 1u);
        __syncthreads();
        MLI_STAMP(ticket, 3);
    }
    __device__ __forceinline__ void row_done(int b) const { sf_add(ctr + kSfMerged + (b / PM) * kSfLine); }
    // the launch's first scan items start beside the projection tiles: their page requests (24 KiB per wave) would queue
    // up in front of the projection's few loads and hold back the one thing everybody waits for
    __device__ __forceinline__ void before_prefetch() const {
        if (first_round) __builtin_amdgcn_s_sleep(64);
    }
    bool first_round;
};

struct StepLogitsGate {
    unsigned* ctr;
    int block;
    unsigned rows;
    unsigned ticket;
    __device__ __forceinline__ void operator()() const {
        MLI_STAMP(ticket, 2);
        if (threadIdx.x == 0) sf_wait<false>(ctr, kSfMerged + block * kSfLine, rows, 2u);
        __syncthreads();
        MLI_STAMP(ticket, 3);
    }
};

struct StepArgs {
    GemmArgs proj;    // kPagedLatest: x rows of the pages . [Wk | Wq | Wv] -> pages, q_output
    GemmArgs logits;  // kPlain, B transposed, argmax epilogue: attention_result . emb_table^T -> row_best
    // scan
    const float* q;
    const void* const* page_table;
    int* lengths;
    float* out;
    float2* ml;
    float* partial;
    unsigned* ctr;    // arrival region: [b] row arrivals of the scan, [kSf...] the counters above
    // head
    int* decoder_result;
    const float* emb_table;
    const float* wpe;
    int B, S, D;
    int ct, ml_per_row, nchunk, tail, slots, grid_rows;
    int n_proj, proj_tiles_x, n_scan, tiles_v, n_blocks;
    int n_results, i_result;
};

// the 32 rows of a block: 8 lanes per row pick the token from the row's (max, index) pairs, thread 0 of the row updates
// the length and the result, the 8 lanes write the next input embedding -- what decoder_finalize_kernel does per wave
__device__ __forceinline__ void step_finalize_rows(const StepArgs& a, int block) {
    const int tid = threadIdx.x;
    const int sub = tid & 7;
    const int b = block * PM + (tid >> 3);
    const bool in_range = b < a.B;
    const int L = in_range ? a.lengths[b] : 0;
    float mv = -FLT_MAX;
    int mi = -1;
    if (in_range && L != 0) {
        for (int t = sub; t < a.tiles_v; t += 8) {
            // 8-byte sc1 load (never from this CU's L1) of a pair another workgroup stored write-through
            typedef unsigned long long __attribute__((address_space(1)))* gu64_ptr;
            const unsigned long long packed = __hip_atomic_load((gu64_ptr)(a.logits.row_best + (int64_t)b * a.tiles_v + t),
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            argmax_take(mv, mi, __uint_as_float((unsigned)packed), (int)(unsigned)(packed >> 32));
        }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
        const float ov = __shfl_xor(mv, off, kWave);
        const int oi = __shfl_xor(mi, off, kWave);
        argmax_take(mv, mi, ov, oi);
    }
    if (!in_range) return;
    if (L == 0) {  // empty slot
        if (sub == 0) a.decoder_result[(int64_t)b * a.n_results + a.i_result] = MLI_EMPTY_ROW_TOKEN_ID;
        return;
    }
    const int tok = mi;
    const bool done = (L + 1 >= a.S) || tok == MLI_EOF_TOKEN_ID;
    float* page = nullptr;
    if (!done) page = const_cast<float*>(reinterpret_cast<const float*>(a.page_table[(int64_t)b * (a.S / kPage) + L / kPage]));
    if (sub == 0) {
        a.decoder_result[(int64_t)b * a.n_results + a.i_result] = tok;
        a.lengths[b] = done ? 0 : L + 1;
    }
    if (done || tok < 0 || page == nullptr) return;
    const float4* e = reinterpret_cast<const float4*>(a.emb_table + (int64_t)tok * a.D);
    const float4* p = reinterpret_cast<const float4*>(a.wpe + (int64_t)L * a.D);
    float4* dst = reinterpret_cast<float4*>(page + page_row_offset(L, a.D, kSegInp));
    for (int i = sub; i < (a.D >> 2); i += 8) {
        const float4 x = e[i], y = p[i];
        dst[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

template <int NJ, bool NT, int TBR>
__global__ __launch_bounds__(kFuThreads, 2) void decode_step_fused_kernel(const StepArgs a) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ unsigned ticket_sh;
    __shared__ int last_sh;
    unsigned* ctr = a.ctr;
#ifdef MLI_STEP_TRACE
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x == 0)
        ticket_sh = __hip_atomic_fetch_add((sf_u32_ptr)(ctr + kSfTicket), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unsigned t = ticket_sh;
#ifdef MLI_STEP_TRACE
    const unsigned t0 = t;
    MLI_NOTE(t0, 0, t_entry);
    MLI_STAMP(t0, 1);
    MLI_NOTE(t0, 5, __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF);
    MLI_NOTE(t0, 6, t < (unsigned)a.n_proj ? 1 : t < (unsigned)(a.n_proj + a.n_scan) ? 2 : 3);
#else
    const unsigned t0 = t;
#endif

    // ---- projection tile ----
    if (t < (unsigned)a.n_proj) {
        const int by = (int)t / a.proj_tiles_x, bx = (int)t % a.proj_tiles_x;
        gemm_panel_tile<kPagedLatest, false, true>(a.proj, bx, by, smem_raw, PanelNoGate{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have left
        __syncthreads();
        if (threadIdx.x == 0) sf_add(ctr + kSfProj + by * kSfLine);
        MLI_STAMP(t0, 4);
        return;
    }
    t -= (unsigned)a.n_proj;

    // ---- scan item: rows fast, full chunks first, the remainders' grid rows last (as the stand-alone grid) ----
    if (t < (unsigned)a.n_scan) {
        // row blocks in order (the block's projection tiles have the lowest tickets, its logits can start while later
        // blocks are still scanned); inside a block rows fast, full chunks first, the remainders' grid rows last
        const int per_block = PM * a.grid_rows;
        const int block = (int)t / per_block, r = (int)t % per_block;
        const int rows_here = min(PM, a.B - block * PM);
        const int b = block * PM + r % rows_here, c = r / rows_here;
        fused_scan_item<ElemF32, NJ, NT, TBR, kFuWaves, false, false>(
            a.q, a.page_table, a.lengths, nullptr, a.out, a.ml, a.partial, a.S, a.D, a.ct, a.ml_per_row, a.nchunk,
            /*direct=*/0, a.tail, a.slots, /*arrivals=*/ctr, b, c, c == 0, a.B, smem_raw,
            StepScanGate{ctr, (unsigned)a.proj_tiles_x, t0, t0 < 512u});
        MLI_STAMP(t0, 4);
        return;
    }
    t -= (unsigned)a.n_scan;

    // ---- logits tile with the argmax epilogue; the block's last tile finalizes its 32 rows ----
    const int block = (int)t / a.tiles_v, tile = (int)t % a.tiles_v;
    if (block >= a.n_blocks) return;  // (the grid is exactly n_proj + n_scan + n_blocks * tiles_v: cannot happen)
    const unsigned rows = (unsigned)min(PM, a.B - block * PM);
    gemm_panel_tile<kPlain, true, true>(a.logits, tile, block, smem_raw, StepLogitsGate{ctr, block, rows, t0});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add((sf_u32_ptr)(ctr + kSfHead + block * kSfLine), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = before + 1u == (unsigned)a.tiles_v;
        // every scan item of these rows holds its length by now, or will in a moment: lengths[b] may change after that
        if (last) sf_wait<false>(ctr, kSfLenRead + block * kSfLine, rows * (unsigned)a.grid_rows, 3u);
        last_sh = last;
    }
    __syncthreads();
    MLI_STAMP(t0, 4);
    if (!last_sh) return;
    step_finalize_rows(a, block);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add((sf_u32_ptr)(ctr + kSfBlocksDone), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = before + 1u == (unsigned)a.n_blocks;
    }
    __syncthreads();
    MLI_STAMP(t0, 7);
    if (!last_sh) return;
    // the launch's last workgroup: every ticket is taken, every counter has been read for the last time
    for (int i = threadIdx.x; i < a.n_blocks; i += kFuThreads) {
        sf_set(ctr + kSfProj + i * kSfLine, 0u);
        sf_set(ctr + kSfLenRead + i * kSfLine, 0u);
        sf_set(ctr + kSfMerged + i * kSfLine, 0u);
        sf_set(ctr + kSfHead + i * kSfLine, 0u);
    }
    if (threadIdx.x == 0) {
        sf_set(ctr + kSfTicket, 0u);
        sf_set(ctr + kSfBlocksDone, 0u);
    }
}

static int g_step_fused = 0;  // mli_tune "step_fused": 1 = small fp32 paged batches run the step as this one launch
void set_step_fused(int v) { g_step_fused = v != 0; }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// 1 = ran, 0 = this batch is not one for the one-launch step (the caller issues the separate launches), else an error
// (+1 if positive).  ws = workspace BODY (the counters sit in front of it), scratch = RowBest pairs of the head.
int launch_decode_step_fused(float* const* page_table, int* lengths, const float* wk, const float* wq, const float* wv,
                             const float* emb_table, const float* wpe, float* q_output, float* attention_result,
                             int* decoder_result, int B, int S, int D, int V, int n_results, int i_result, void* ws,
                             size_t ws_bytes, void* scratch, size_t scratch_bytes, hipStream_t st) {
    if (!g_step_fused || ws == nullptr || scratch == nullptr) return 0;
    if (B <= 0 || V <= 0 || B > kSfMaxBlocks * PM || D % 4 != 0 || S % kPage != 0) return 0;
    const bool vec4 = aligned16(wk) && aligned16(wq) && aligned16(wv) && aligned16(emb_table) && aligned16(wpe) &&
                      aligned16(attention_result);
    // the small-batch shapes only: both products are the panel kernel's, the scan is the chunked grid's
    if (!gemm_panel_wanted(B, 3 * D, D, vec4) || !gemm_panel_wanted(B, V, D, vec4)) return 0;
    ChunkedScanPlan plan;
    if (!plan_chunked_scan_f32(B, S, D, ws_bytes, &plan)) return 0;
    const int tiles_v = ceil_div_i(V, PN);
    if (scratch_bytes < (size_t)B * tiles_v * sizeof(RowBest)) return 0;

    StepArgs a{};
    a.proj.w[0] = wk; a.proj.w[1] = wq; a.proj.w[2] = wv; a.proj.n_out = 3;
    a.proj.out_id[0] = 0; a.proj.out_id[1] = 1; a.proj.out_id[2] = 2;
    a.proj.M = B; a.proj.N = D; a.proj.K = D;
    a.proj.page_table = page_table; a.proj.q_output = q_output; a.proj.lengths = lengths;
    a.proj.B = B; a.proj.S = S;
    a.logits.w[0] = emb_table; a.logits.n_out = 1; a.logits.out_id[0] = 1;
    a.logits.M = B; a.logits.N = V; a.logits.K = D;
    a.logits.a_plain = attention_result; a.logits.c_plain = nullptr; a.logits.lda = D; a.logits.ldc = V;
    a.logits.row_best = reinterpret_cast<RowBest*>(scratch);
    a.q = q_output;
    a.page_table = reinterpret_cast<const void* const*>(page_table);
    a.lengths = lengths;
    a.out = attention_result;
    a.ml = reinterpret_cast<float2*>(ws);
    a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + plan.stats_bytes);
    a.ctr = ws_arrivals(ws);
    a.decoder_result = decoder_result;
    a.emb_table = emb_table;
    a.wpe = wpe;
    a.B = B; a.S = S; a.D = D;
    a.ct = plan.ct; a.ml_per_row = plan.ml_per_row; a.nchunk = plan.nchunk; a.tail = plan.tail; a.slots = plan.slots;
    a.grid_rows = plan.grid_rows;
    a.n_blocks = ceil_div_i(B, PM);
    a.proj_tiles_x = 3 * ceil_div_i(D, PN);
    a.n_proj = a.n_blocks * a.proj_tiles_x;
    a.n_scan = B * plan.grid_rows;
    a.tiles_v = tiles_v;
    a.n_results = n_results; a.i_result = i_result;
    const size_t smem = plan.smem > kPanelSmem ? plan.smem : kPanelSmem;
    const int64_t total = (int64_t)a.n_proj + a.n_scan + (int64_t)a.n_blocks * tiles_v;
    if (total > (1 << 30)) return 0;

#define MLI_SF_LAUNCH(NJ, NT, TBR)                                                                                    \
    do {                                                                                                               \
        auto kern = decode_step_fused_kernel<NJ, NT, TBR>;                                                             \
        static std::atomic<unsigned long long> opted_in{0};                                                            \
        int device = 0;                                                                                                \
        (void)hipGetDevice(&device);                                                                                   \
        const unsigned long long bit = 1ull << (device & 63);                                                          \
        if (!(opted_in.load(std::memory_order_relaxed) & bit)) { /* > 64 KiB of dynamic LDS: opt in once per device */ \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            if (e != hipSuccess) return (int)e + 1;                                                                    \
            opted_in.fetch_or(bit, std::memory_order_relaxed);                                                         \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(kFuThreads), smem, st, a);                                \
    } while (0)
    if (plan.nj == 1) {
        if (plan.nt) MLI_SF_LAUNCH(1, true, 8);
        else MLI_SF_LAUNCH(1, false, 8);
    } else {
        if (plan.nt) MLI_SF_LAUNCH(2, true, 4);
        else MLI_SF_LAUNCH(2, false, 4);
    }
#undef MLI_SF_LAUNCH
    const int rc = launch_status();
    return rc ? (rc > 0 ? rc + 1 : rc) : 1;
}

}  // namespace mli

#ifdef MLI_STEP_TRACE
extern "C" int mli_debug_step_trace(unsigned long long* host, int n_slots) {
    if (n_slots > mli::kStepTraceSlots) n_slots = mli::kStepTraceSlots;
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mli::mli_step_trace), (size_t)n_slots * 8 * sizeof(unsigned long long));
}
extern "C" int mli_debug_step_trace_clear(void) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(mli::mli_step_trace));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * mli::kStepTraceSlots * 8);
}
#endif

// diagnostic: the error word of the one-launch step (0 = every wait of every launch so far ended normally; 1 / 2 / 3 =
// a scan item / a logits tile / a finalizer gave up waiting).  Synchronises the device.
extern "C" int mli_debug_step_fused_error(void* workspace, size_t workspace_bytes, unsigned* code_out) {
    if (workspace == nullptr || workspace_bytes <= mli::kArrivalRegionBytes || code_out == nullptr) return MLI_ERR_BAD_ARG;
    return (int)hipMemcpy(code_out, reinterpret_cast<unsigned*>(workspace) + mli::kSfError, sizeof(unsigned), hipMemcpyDeviceToHost);
}
