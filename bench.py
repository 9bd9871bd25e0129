#!/usr/bin/env python3
"""Decode-step benchmark for the MI355X decode attention path.

    python bench.py --gpus 1 --steps 50 --warmup 10            # one GPU
    python bench.py --gpus N --steps K --warmup W               # N GPUs: starts N rank processes itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W    # the same, launched by the caller (RCCL)

One step = one decode step of the continuous batch on synthetic state already resident in HBM:
attention block (projection GEMM with page gather/scatter -> q.K^T -> masked softmax -> softmax.V)
followed by the greedy decoder head, i.e. what PagedAttentionInferenceModel::forward does per
iteration with n_new_items = 0 (reference src/inference_model.cpp:52-82).  Every rank owns an
independent shard of the batch rows (weak scaling); the only cross-GPU traffic is the all-gather of
the generated token ids.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

# dmabuf IPC is the only IPC flavour the host driver supports: RCCL / cross-process device memory need it.  Set in the
# rank itself, before anything initialises the GPU runtime -- ranks started by an outer torch.distributed.run never pass
# through self_launch() below.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from min_llm_inference_amd import ops  # noqa: E402

PAGE = 16
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (layout, rows per GPU, emb_dim, max_seq) -- BASELINE.json configs[1..3]
    "c2": ("naive", 256, 256, 1024),
    "c3": ("paged", 256, 256, 1024),
    "c4": ("paged", 1024, 512, 4096),
    # the reference's own profiling shape (tests/paged_for_profile.cpp:11-23) as a fixed-state decode step
    "e1": ("paged", 1024, 2048, 128),
}
N_VOCAB = 1024
ELEM_NAMES = {"f32": "fp32", "bf16": "bf16 pages and weights", "fp8": "fp8 (OCP e4m3) pages, bf16 weights"}
SCAN_LEAN = ("fused_decode_scan, lean (q.K^T + online softmax + softmax.V + in-kernel merge, one visit per page; "
             "fused_decode_stream_kernel = equal page shares where the batch fills the chip, else fused_decode_scan_kernel)")
SCAN_FULL = "fused_decode_scan, materialising (raw scores written; merged by fused_decode_combine)"
SCAN_NAIVE = ("naive_decode_scan, lean (contiguous caches: 256-token chunks scored from the K^T tile, probabilities in LDS, "
              "accumulated over the V tile, rows merged in-kernel)")


class Workload:
    """Synthetic decode state for one GPU (weights U(-1,1)/sqrt(D), lengths U[S/4, 3S/4],
    pages drawn from a shuffled pool)."""

    def __init__(self, name, dev, seed, headroom, dtype="f32"):
        self.name = name
        self.layout, self.B, self.D, self.S = WORKLOADS[name]
        self.dtype = dtype
        self.esize = {"f32": 4, "bf16": 2, "fp8": 1}[dtype]  # bytes per page element (K, V, x)
        self.wsize = 4 if dtype == "f32" else 2              # bytes per weight element (fp8 pages: bf16 weights)
        self.elem = {"f32": ops.ELEM_F32, "bf16": ops.ELEM_BF16, "fp8": ops.ELEM_FP8}[dtype]
        if dtype != "f32" and self.layout != "paged":
            raise SystemExit("bf16 / fp8 are implemented for the paged layout (BASELINE config 4)")
        B, D, S = self.B, self.D, self.S
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        rng = np.random.default_rng(seed)
        self.dev = dev
        hi = min(3 * S // 4, S - headroom - 3)  # every row must stay below max_seq for the whole run
        if hi < S // 4:
            raise SystemExit(f"workload {name}: {headroom - 8} steps do not fit into max_seq {S}; use fewer --steps")
        lengths = rng.integers(S // 4, hi + 1, size=B).astype(np.int32)
        if os.environ.get("MLI_BENCH_LENGTHS") == "uniform":  # diagnostic: no length variation between rows
            lengths[:] = S // 2
        assert int(lengths.max()) + headroom + 2 < S
        self.lengths_host = lengths
        self.lengths = torch.from_numpy(lengths).to(dev)
        self.lengths0 = self.lengths.clone()

        def u(*shape, scale=1.0):
            return (torch.rand(*shape, device=dev, generator=g) * 2 - 1) * scale

        sc = 1.0 / np.sqrt(D)
        self.wk, self.wq, self.wv = (u(D, D, scale=sc) for _ in range(3))
        if dtype != "f32":
            self.wk, self.wq, self.wv = (w.to(torch.bfloat16) for w in (self.wk, self.wq, self.wv))
        self.emb_table = u(N_VOCAB, D)
        self.emb_table[ops.EOF_TOKEN_ID] = 0  # EOF never wins the argmax: the batch stays full while timing
        self.wpe = u(S, D)
        self.q_output = torch.zeros(B, D, device=dev)
        self.qkt_output = torch.zeros(B, S, device=dev)
        self.attention_result = torch.zeros(B, D, device=dev)
        self.emb_score = torch.zeros(B, N_VOCAB, device=dev)
        self.decoder_result = torch.full((B, 1), -1, dtype=torch.int32, device=dev)
        self.new_idx = torch.zeros(B, dtype=torch.int32, device=dev)
        if self.layout == "paged":
            W = S // PAGE
            per_row = [-(-(int(L) + headroom + 2) // PAGE) for L in lengths]
            total = sum(per_row)
            block = PAGE * 3 * D
            self.pool = torch.empty(total * block, device=dev,
                                    dtype={"f32": torch.float32, "bf16": torch.bfloat16, "fp8": torch.uint8}[dtype])
            chunk = 1 << 28
            for o in range(0, self.pool.numel(), chunk):  # K/V/x contents: U(-1,1)
                n = min(chunk, self.pool.numel() - o)
                if dtype == "fp8":   # OCP e4m3 codes, by the conversion the page kernels use
                    ops.f32_to_fp8(u(n), self.pool[o:o + n])
                else:
                    self.pool[o:o + n] = u(n).to(self.pool.dtype)
            order = rng.permutation(total)
            if os.environ.get("MLI_BENCH_POOL_ORDER") == "linear":  # diagnostic: pages handed out in address order
                order = np.arange(total)
            table = np.zeros((B, W), np.int64)
            self.page_ids = np.full((B, W), -1, np.int64)   # which pool block backs (row, page): tests index the pool with it
            cur = 0
            base = self.pool.data_ptr()
            for b in range(B):
                table[b, :per_row[b]] = base + self.esize * block * order[cur:cur + per_row[b]].astype(np.int64)
                self.page_ids[b, :per_row[b]] = order[cur:cur + per_row[b]]
                cur += per_row[b]
            self.page_table = torch.from_numpy(table).to(dev)
            self.kv_bytes_resident = self.pool.numel() * self.esize
        else:
            self.inp_embedding = u(B, S, D)
            self.kt_cache = u(B, D, S)
            self.v_cache = u(B, S, D)
        ops.workspace_for(B, S, D, dev)  # allocate scratch outside the timed region
        torch.cuda.synchronize()

    def attention(self):
        if self.dtype == "fp8":
            raise SystemExit("fp8 pages have the lean composition only")
        if self.dtype == "bf16":
            ops.paged_attention_bf16(self.page_table, self.lengths, self.wk, self.wq, self.wv, self.new_idx,
                                     self.q_output, self.qkt_output, self.attention_result, 0, self.S)
        elif self.layout == "paged":
            ops.paged_attention(self.page_table, self.lengths, self.wk, self.wq, self.wv, self.new_idx, self.q_output,
                                self.qkt_output, self.attention_result, 0, self.S)
        else:
            ops.inference_self_attention(self.inp_embedding, self.lengths, self.wk, self.wq, self.wv, self.new_idx,
                                         self.kt_cache, self.v_cache, self.q_output, self.qkt_output,
                                         self.attention_result, 0)

    def decoder(self):
        if self.dtype == "bf16":
            ops.launch_paged_attention_decoder_multi_rounds_bf16(self.attention_result, self.emb_table, self.emb_score,
                                                                 self.wpe, self.page_table, self.lengths,
                                                                 self.decoder_result, 0)
        elif self.layout == "paged":
            ops.launch_paged_attention_decoder_multi_rounds(self.attention_result, self.emb_table, self.emb_score,
                                                            self.wpe, self.page_table, self.lengths,
                                                            self.decoder_result, 0)
        else:
            ops.launch_decoder(self.attention_result, self.emb_table, self.emb_score, self.wpe, self.inp_embedding,
                               self.lengths, self.decoder_result.view(-1))

    def step(self):
        self.attention()
        self.decoder()

    def rows(self, lo, hi):
        """A view of batch rows [lo, hi): the same weights, pool and tables, row-indexed state sliced (rows are
        independent in every kernel, so a slice is a complete smaller batch -- what a micro-batch or a rank owns)."""
        import copy
        v = copy.copy(self)
        v.B = hi - lo
        for name in ("lengths", "lengths0", "q_output", "qkt_output", "attention_result", "emb_score", "decoder_result",
                     "new_idx", "page_table", "inp_embedding", "kt_cache", "v_cache"):
            if hasattr(self, name):
                setattr(v, name, getattr(self, name)[lo:hi])
        v.lengths_host = self.lengths_host[lo:hi]
        return v

    # ---- what the layers / engines run: no scores or probabilities materialised, no emb_score ------------------
    def lean_attention(self):
        if self.layout == "paged":
            ops.paged_attention_lean(self.page_table, self.lengths, self.wk, self.wq, self.wv, self.new_idx,
                                     self.q_output, self.attention_result, 0, self.S, elem=self.elem)
        else:  # the contiguous layout keeps K transposed: its two passes need the score buffer between them
            self.attention()

    def fused_decoder(self):
        if self.layout == "paged":
            ops.paged_decoder_fused(self.attention_result, self.emb_table, self.wpe, self.page_table, self.lengths,
                                    self.decoder_result, 0, self.elem)
        else:
            ops.decoder_fused(self.attention_result, self.emb_table, self.wpe, self.inp_embedding, self.lengths,
                              self.decoder_result.view(-1))

    def lean_step(self):
        """One decode step through the one-call entry point (mli_paged_decode_step / mli_decode_step): the launches of
        lean_attention() + fused_decoder(), issued from C like the C++ layers issue them -- the arguments are marshalled
        once per (state, stream), so the timed loop measures the step, not the Python front end."""
        stream = torch.cuda.current_stream().cuda_stream
        key = (stream, self.decoder_result.data_ptr(), self.lengths.data_ptr())
        cache = self.__dict__.setdefault("_step_cache", {})  # (the token gather alternates between two result buffers)
        if key not in cache:
            lib = ops.load_library()
            ws, need = ops.workspace_for(self.B, self.S, self.D, self.dev)
            sc, sc_need = ops.decoder_scratch_for(self.B, N_VOCAB, self.dev)
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            if self.layout == "paged":
                fn = lib.mli_paged_decode_step
                args = (p(self.page_table), p(self.lengths), p(self.wk), p(self.wq), p(self.wv),
                                   p(self.emb_table), p(self.wpe), p(self.q_output), p(self.attention_result),
                                   p(self.decoder_result), self.B, self.S, self.D, N_VOCAB, 1, 0,
                                   self.elem, p(ws), need, p(sc), sc_need, ctypes.c_void_p(stream))
            else:
                fn = lib.mli_decode_step
                args = (p(self.inp_embedding), p(self.lengths), p(self.wk), p(self.wq), p(self.wv),
                                   p(self.emb_table), p(self.wpe), p(self.kt_cache), p(self.v_cache), p(self.q_output),
                                   p(self.qkt_output), p(self.attention_result), p(self.decoder_result), self.B, self.S,
                                   self.D, N_VOCAB, p(ws), need, p(sc), sc_need, ctypes.c_void_p(stream))
            cache[key] = (fn, args, ws, sc)
        fn, args = cache[key][:2]
        rc = fn(*args)
        if rc != 0:
            raise ops.MliError(f"decode step returned {rc}")

    # ---- the individual launches, for the roofline pass ------------------------------------
    def kernels(self, lean=True):
        """name -> callable for every launch of one decode step, in order.  lean = what the layers run."""
        w = self
        bf = self.elem
        if self.layout == "paged":
            k = {"get_latest_k_q_v_paged (MFMA gather-GEMM-scatter)": lambda: ops.get_latest_k_q_v_paged_lean(
                w.page_table, w.lengths, w.wk, w.wq, w.wv, w.q_output, w.S, elem=w.elem)}
            if lean:
                k[SCAN_LEAN] = lambda: ops.decode_scan_paged(w.q_output, w.page_table, w.lengths, None, w.attention_result,
                                                             bf, phases=7, n_sequence=w.S)
                k["decoder head (logits GEMM with argmax epilogue + finalize)"] = w.fused_decoder
            else:
                k[SCAN_FULL] = lambda: ops.decode_scan_paged(w.q_output, w.page_table, w.lengths, w.qkt_output,
                                                             w.attention_result, bf, phases=1)
                k["fused_decode_combine (merge + probabilities)"] = lambda: ops.decode_scan_paged(
                    w.q_output, w.page_table, w.lengths, w.qkt_output, w.attention_result, bf, phases=2)
                k["decoder head (logits GEMM + argmax kernel, emb_score materialised)"] = w.decoder
            return k
        if lean:
            return {
                "get_latest_kt_q_v (MFMA GEMM)": lambda: ops.launch_get_latest_kt_q_v(
                    w.inp_embedding, w.lengths, w.wk, w.wq, w.wv, w.kt_cache, w.v_cache, w.q_output),
                SCAN_NAIVE: lambda: ops.decode_scan_contiguous(w.q_output, w.kt_cache, w.v_cache, w.lengths,
                                                               w.attention_result),
                "decoder head (logits GEMM with argmax epilogue + finalize)": w.fused_decoder,
            }
        return {
            "get_latest_kt_q_v (MFMA GEMM)": lambda: ops.launch_get_latest_kt_q_v(
                w.inp_embedding, w.lengths, w.wk, w.wq, w.wv, w.kt_cache, w.v_cache, w.q_output),
            "qkt": lambda: ops.launch_qkt(w.q_output, w.kt_cache, w.lengths, w.qkt_output),
            "softmax_in_place_with_lengths": lambda: ops.launch_softmax_in_place_with_lengths(w.qkt_output, w.lengths),
            "softmax_v": lambda: ops.launch_softmax_v(w.qkt_output, w.v_cache, w.attention_result, w.lengths),
            "decoder head": w.decoder,
        }

    def algorithmic_bytes(self, lengths):
        """Per-launch algorithmic HBM bytes (DESIGN.md 'Roofline accounting'); e = bytes per KV/weight element."""
        L = lengths.astype(np.int64)
        B, D, e, we = self.B, self.D, self.esize, self.wsize
        live = int((L > 0).sum())
        ptrs = int((8 * -(-L // PAGE)).sum()) if self.layout == "paged" else 0
        kv_one = int(L.sum()) * D * e
        qkt = kv_one + live * D * 4 + int(L.sum()) * 4 + ptrs + B * 4       # K + q in, scores out
        sv = kv_one + int(L.sum()) * 4 + live * D * 4 + ptrs + B * 4        # V + probs in, result out
        latest = live * D * (3 * e + 4) + 3 * D * D * we + B * 4 + (8 * live if self.layout == "paged" else 0)
        step = (2 * kv_one + live * (3 * D * e + D * 4) + 3 * D * D * we + B * 4 + ptrs)
        scan = 2 * kv_one + live * D * 4 + int(L.sum()) * 4 + ptrs + B * 4   # K + V + q in, raw scores out
        scan_lean = 2 * kv_one + 2 * live * D * 4 + ptrs + B * 4             # K + V + q in, attention_result out
        return {"qkt": qkt, "softmax_v": sv, "scan": scan, "scan_lean": scan_lean, "get_latest": latest, "step": step}


def time_kernel(fn, reps, batch=1):
    """Average duration of fn's launches in ms, HIP events on the stream they are launched on (torch's current
    stream).  batch > 1: `batch` calls are captured into one hipGraph and replayed, so that a launch of a few
    microseconds is not timed at the rate Python can issue it."""
    fn()
    st = torch.cuda.current_stream()
    st.synchronize()
    g = None
    if batch > 1:
        g = ops.StepGraph(lambda: [fn() for _ in range(batch)])
        run, n, per = g.launch, max(1, reps // batch), batch
    else:
        run, n, per = fn, reps, 1
    run()
    st.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        run()
    e1.record(st)
    st.synchronize()
    if g is not None:
        g.close()
    return e0.elapsed_time(e1) / (n * per)


def measure_read_gbs(dev):
    """The box's streaming-read rate (mli_stream_read: every wave streams its own contiguous region with 16-byte
    non-temporal lane loads, 16 KiB in flight) -- the ceiling a read-only scan can be held against.  (A device COPY, which
    earlier rounds reported here, is not a ceiling for a read stream: its writes cost more than its reads.)"""
    n = 1 << 30  # 4 GiB of floats
    a = torch.empty(n, device=dev)
    a.uniform_()
    sink = torch.zeros(64, device=dev)
    ms = time_kernel(lambda: ops.stream_read(a, sink), 10)
    del a
    return n * 4 / (ms * 1e-3) / 1e9


def cpu_baseline(wl, budget_s=12.0):
    """The CPU oracle (single-threaded port of the reference's host functions) on a bounded row sample of the
    same workload, contiguous layout (the reference has no CPU paged path), decode step only (n_new = 0)."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    D, S = wl.D, wl.S
    rng = np.random.default_rng(1)
    n = int(min(wl.B, 4e9 // (3 * S * D * 4)))  # row sample bounded by ~4 GB of host memory
    L = wl.lengths_host[:n].copy()
    inp = rng.random((n, S, D), dtype=np.float32)
    kt = rng.random((n, D, S), dtype=np.float32)
    v = rng.random((n, S, D), dtype=np.float32)
    w = [((rng.random((D, D), dtype=np.float32) * 2 - 1) / np.sqrt(D)).astype(np.float32) for _ in range(3)]
    q = np.zeros((n, D), np.float32)
    s = np.zeros((n, S), np.float32)
    o = np.zeros((n, D), np.float32)
    idx = np.zeros((n,), np.int32)
    live = int((L > 0).sum())

    def rows(lo, hi):  # one decode step over rows [lo, hi): contiguous row slices are contiguous arrays
        oracle.self_attention_inference_host(inp[lo:hi], L[lo:hi], w[0], w[1], w[2], idx[lo:hi], kt[lo:hi], v[lo:hi],
                                             q[lo:hi], s[lo:hi], o[lo:hi], 0)

    total_t, reps = 0.0, 0
    while total_t < budget_s and reps < 50:      # repeat the decode step over the sample until ~budget_s of CPU work
        t0 = time.perf_counter()
        rows(0, n)
        total_t += time.perf_counter() - t0
        reps += 1
    # the same step with the rows dealt to every host core (the reference's CPU path is single-threaded; rows are
    # independent, so this is the obvious parallel form of it -- SURVEY 8(d)(ii)); ctypes calls release the GIL
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nthr = max(1, min(cores, n, int(os.environ.get("MLI_CPU_THREADS", "16"))))  # a one-GPU box's CPU share is 16
    cuts = [n * i // nthr for i in range(nthr + 1)]
    mt_t, mt_reps = 0.0, 0
    with ThreadPoolExecutor(nthr) as pool:
        while mt_t < budget_s / 2 and mt_reps < 200:
            t0 = time.perf_counter()
            list(pool.map(lambda i: rows(cuts[i], cuts[i + 1]), range(nthr)))
            mt_t += time.perf_counter() - t0
            mt_reps += 1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": live * reps / total_t, "unit": "tokens/s", "cores": 1, "kind": "port", "host_cpu": cpu_model,
            "host_cores_visible": cores,
            "sample": f"{reps} decode steps over {n} of {wl.B} rows of the same workload shape (contiguous layout, "
                      f"n_new=0, fp32 -- the reference's CPU path has no paged or bf16 form), {total_t:.1f} s "
                      f"single-threaded oracle_cpu.c",
            "all_cores": {"value": live * mt_reps / mt_t, "unit": "tokens/s", "cores": nthr,
                          "sample": f"{mt_reps} steps over the same rows split across {nthr} threads, {mt_t:.1f} s"}}


MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "fp8": 2500.0}  # (fp8 pages: the projection multiplies in bf16)  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


def gemm_report(wl, lengths, times):
    """MFMA utilisation of the decode projection x[B,D].[Wk|Wq|Wv] (gather-GEMM-scatter kernel)."""
    live = int((lengths > 0).sum())
    flops = 2.0 * live * wl.D * 3 * wl.D
    ms = [v for k, v in times.items() if k.startswith("get_latest")][0]
    tf = flops / (ms * 1e-3) / 1e12
    peak = MFMA_PEAK_TFLOPS[wl.dtype]
    return {"bound": "mfma", "flops_per_launch": flops, "avg_launch_ms": ms, "achieved": tf, "peak": peak,
            "unit": "TFLOP/s", "frac": tf / peak,
            "note": (f"{flops / 1e9:.1f} GFLOP per launch at B={wl.B}, D={wl.D}" +
                     (": latency-bound, 2 % of the step; the same kernels at D=2048 (bench.py --workload e1) reach "
                      "110-145 TFLOP/s fp32 / 650 TFLOP/s bf16 (DESIGN.md 3.5)" if wl.D <= 512 else ""))}


def large_gemm_report(dev):
    """The same projection kernels where they are not latency-bound: x[1024, 2048].[Wk|Wq|Wv] (the reference's
    profiling shape, bench.py --workload e1), both element types, measured here so that the MFMA figure in the JSON
    is not only the 1.6-GFLOP launch of config 4."""
    out = {"shape": "B=1024, D=2048 (25.8 GFLOP per launch)", "unit": "TFLOP/s"}
    for dt in ("f32", "bf16"):
        try:
            w = Workload("e1", dev, 0xE1, headroom=8, dtype=dt)
            fn = [v for k, v in w.kernels().items() if k.startswith("get_latest")][0]
            # the workload's setup leaves the GPU idle for a while and a 3 ms measurement would be taken on the clock ramp
            # (tools/gemm_bf16_probe.py: the first 200 launches run 3-10 % slower than the next): warm up first, take the median
            time_kernel(fn, 300, batch=10)
            ms = sorted(time_kernel(fn, 200, batch=10) for _ in range(3))[1]
            tf = 2.0 * w.B * w.D * 3 * w.D / (ms * 1e-3) / 1e12
            out[dt] = {"avg_launch_ms": ms, "achieved": tf, "peak": MFMA_PEAK_TFLOPS[dt], "frac": tf / MFMA_PEAK_TFLOPS[dt]}
            del w
            torch.cuda.empty_cache()
        except Exception as e:  # never let the side measurement take the bench line down
            out[dt] = {"error": str(e)[:200]}
    return out


def pmc_traffic(workload, which, layout, dtype="f32"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary of this command
    (profiles/pmc_<workload>.json, written by tools/pmc_summary.py; FETCH_SIZE x2 + WRITE_SIZE).  bench.py cannot
    run the counter passes on itself, so this is null when no summary for the workload is committed."""
    path = os.path.join(ROOT, "profiles", f"pmc_{workload}{'' if dtype == 'f32' else '_' + dtype}.json")
    if not os.path.exists(path):
        return None, None, None
    summary = json.load(open(path))
    kernels = summary["kernels"]
    run_bytes = (summary.get("counter_run") or {}).get("algorithmic_bytes_per_launch") if which == "scan_lean" else None
    needle = {"qkt": "qkt_", "softmax_v": "softmax_v_partial", "scan": "fused_decode_scan", "scan_lean": "fused_decode_s"}[which]
    # the chunked scan kernel's template arguments end in "..., SCORES, RPI>": SCORES says whether it writes the raw scores
    # (the materialising form); the equal-shares kernel (fused_decode_stream_kernel) exists in the lean form only
    import re

    def scores_flag(name):
        m = re.search(r",\s*(true|false)(?:,\s*\d+)?>\s*$", name)
        return m.group(1) if m else None

    if which == "scan_lean":
        hits = [v["traffic_bytes"] for k, v in kernels.items() if "fused_decode_stream" in k] or \
               [v["traffic_bytes"] for k, v in kernels.items() if "fused_decode_scan" in k and scores_flag(k) == "false"] or \
               [v["traffic_bytes"] for k, v in kernels.items() if "naive_decode_scan" in k]
    elif which == "scan":
        hits = [v["traffic_bytes"] for k, v in kernels.items() if needle in k and scores_flag(k) == "true"]
    else:
        hits = [v["traffic_bytes"] for k, v in kernels.items() if needle in k]
    return (hits[0], os.path.relpath(path, ROOT), run_bytes) if hits else (None, None, None)


def run_engine_mode(args, rank, world, dev):
    """Engine-level throughput on the reference's profiling workload (tests/paged_for_profile.cpp:11-23):
    B=1024 slots, S=128, D=2048, V=1024, 4096 pages, 2048 items with prompt length U[1,64], one forward round.
    tokens/s = ThroughputCounter (tokens appended / wall time, host scheduling and copies included) -- the
    figure README.md:54-82 quotes (123 284 tok/s on an unnamed NVIDIA GPU).  Each rank runs an independent
    replica of the workload on its own GPU."""
    from min_llm_inference_amd import engine as eng
    ops.load_library().mli_engine_set_lean_layers(0 if args.reference_launch_sequence else 1)
    ops.load_library().mli_engine_set_step_graphs(1 if args.step_graphs else 0)
    _, B, D, S = WORKLOADS[args.workload if args.workload != "c4" or args.engine_shape else "e1"]
    V = N_VOCAB
    rng = np.random.default_rng(0x5EED0100 + rank)

    def u(*shape, scale=1.0):
        return ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(scale)).astype(np.float32)

    emb = u(V, D)
    emb[ops.EOF_TOKEN_ID] *= 1.0001  # the reference scales the EOF row the same way
    kind = {"paged": eng.PAGED, "paged_gemm": eng.PAGED_GEMM, "paged_bf16": eng.PAGED_BF16,
            "paged_fp8": eng.PAGED_FP8}[args.engine_kind]
    # page pool: the reference gives 4 pages per slot at S = 128, i.e. half of the worst case B * S / 16 -- rows
    # outgrow it, so page growth and preemption are part of the measured run (mode E)
    n_blocks = B * S // 32
    R = max(1, args.engine_replicas)
    assert B % R == 0
    weights = (emb, u(S, D), u(D, D, scale=1 / np.sqrt(D)), u(D, D, scale=1 / np.sqrt(D)), u(D, D, scale=1 / np.sqrt(D)))
    items = [(i, rng.integers(0, ops.EOF_TOKEN_ID, size=int(rng.integers(1, 65)))) for i in range(2 * B)]

    def one_run():
        engines = []
        for r in range(R):  # R engines of B / R slots each share the GPU (private streams, one host thread each)
            e = eng.Engine(kind, B // R, S, D, V, *weights, n_blocks=n_blocks // R, n_forward_rounds=1, device=dev.index,
                           reference_length_reset_quirk=args.reference_quirk)
            if R > 1 or args.step_graphs:
                e.use_private_stream()
            if args.pipelined or args.sequential_loop:
                e.set_pipelined(not args.sequential_loop)   # default: the engine picks the pipelined loop where it applies
            for i, toks in items[r::R]:
                e.add_item(i, toks)
            engines.append(e)
        try:
            if R == 1:
                st = engines[0].run()
                assert st.finished == 2 * B
                return types.SimpleNamespace(total_tokens=st.total_tokens, seconds=st.seconds, iterations=st.iterations)
            import threading
            stats = [None] * R
            def drive(r):
                stats[r] = engines[r].run()
            threads = [threading.Thread(target=drive, args=(r,)) for r in range(R)]
            t0 = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            wall = time.perf_counter() - t0
            assert sum(s.finished for s in stats) == 2 * B
            # whole-GPU totals: tokens of all engines over the wall time of the slowest
            return types.SimpleNamespace(total_tokens=sum(s.total_tokens for s in stats),
                                         seconds=max(max(s.seconds for s in stats), wall),
                                         iterations=max(s.iterations for s in stats))
        finally:
            for e in engines:
                e.close()

    import types
    runs = [one_run() for _ in range(max(1, args.engine_repeats))]
    rates = sorted((r.total_tokens / r.seconds, i) for i, r in enumerate(runs))
    st = runs[rates[len(rates) // 2][1]]                        # the median run is the one reported
    st.repeat = {"runs": len(runs), "median": rates[len(rates) // 2][0], "min": rates[0][0], "max": rates[-1][0],
                 "all": [r.total_tokens / r.seconds for r in runs], "unit": "tokens/s"}
    return st, (B, S, D, V, n_blocks)


def launcher_command(n_ranks, argv, port):
    """The command `python bench.py --gpus N` runs on behalf of the caller: the same launch line the driver uses
    (one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(n_ranks, argv):
    """Start N fresh rank processes (children of this one, never an exec: this process stays a plain launcher that
    has not initialised the GPU), relay what they print -- rank 0's JSON line -- and return their exit code."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(launcher_command(n_ranks, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:  # stderr passes straight through
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c4")
    ap.add_argument("--dtype", choices=["auto", "f32", "bf16", "fp8"], default="auto",
                    help="KV page / weight element type (accumulation, q, scores and outputs are always fp32); "
                         "auto = what BASELINE.json names for the workload: bf16 for c4, fp32 for c2/c3; fp8 = the opt-in "
                         "extension (OCP e4m3 pages, bf16 weights; lean composition only) -- never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` block (short runs of the other BASELINE configurations and the engine workload)")
    ap.add_argument("--config-steps", type=int, default=50, help="timed steps of every run in the `configs` block")
    ap.add_argument("--materialising", action="store_true",
                    help="time the reference's launch sequence (scores, probabilities and emb_score written) as the "
                         "headline instead of the lean composition the layers run")
    ap.add_argument("--mode", choices=["step", "engine"], default="step",
                    help="step: kernel-level decode step (default); engine: the reference's profiling workload end to end")
    ap.add_argument("--engine-kind", choices=["paged", "paged_gemm", "paged_bf16", "paged_fp8"], default="paged_gemm",
                    help="paged_bf16 = extension: bf16 pages and weights (BASELINE config 4 dtype); paged_fp8 = opt-in "
                         "extension: fp8 (OCP e4m3) pages, bf16 weights")
    ap.add_argument("--pipelined", action="store_true",
                    help="engine mode: insist on the pipelined loop (host one step behind the GPU; per-slot device "
                         "updates) -- it is the default wherever it applies")
    ap.add_argument("--step-graphs", action="store_true",
                    help="engine mode: private stream + decode forwards replayed from a hipGraph (one host call each)")
    ap.add_argument("--sequential-loop", action="store_true",
                    help="engine mode: the reference's sequential loop order (forward, result, pages, insert)")
    ap.add_argument("--engine-replicas", type=int, default=1,
                    help="engine mode: split the slots over this many engines on the same GPU (private streams, one "
                         "host thread each) so one engine's host bookkeeping overlaps the other's kernels")
    ap.add_argument("--engine-repeats", type=int, default=5,
                    help="engine mode: runs of the workload (a fresh engine each, same items); the median run is reported")
    ap.add_argument("--repeats", type=int, default=5,
                    help="step mode: timed regions of exactly --steps steps each (same state); the median region is `value`")
    ap.add_argument("--engine-shape", action="store_true",
                    help="engine mode: run the --workload shape (e.g. c4) instead of the reference's profiling shape e1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU work: start the ranks, rendezvous over gloo, print one line (tests the launch path on CPU)")
    ap.add_argument("--reference-launch-sequence", action="store_true",
                    help="engine mode: the layers issue the reference's launch sequence (encoder, fill, latest, scan + "
                         "combine, logits, argmax) instead of the lean compositions; same tokens")
    ap.add_argument("--reference-quirk", action="store_true",
                    help="engine mode: reproduce the reference's stale-length upload (DESIGN.md deviation 2)")
    args = ap.parse_args()
    if args.dtype == "auto":
        args.dtype = "bf16" if args.workload == "c4" else "f32"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (nothing has touched the GPU yet)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (plain `python bench.py --gpus N` "
                         "starts the ranks itself; under torch.distributed.run pass --nproc-per-node N)")
    if args.rehearse_launch:
        # launcher -> N ranks -> rendezvous -> collective -> ONE line from rank 0, with no GPU work at all: what the
        # CPU test of the self-launch path runs (tests/test_bench_launcher.py).  Never a measurement.
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = torch.tensor([1.0])
        if world > 1:
            dist.all_reduce(ranks)
        if rank == 0:
            print(json.dumps({"rehearsal": True, "n_gpus": world, "ranks_seen": int(ranks.item()), "value": None}),
                  flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and args.backend == "nccl":
        raise SystemExit(f"rank {local_rank} has no GPU ({n_dev} visible): one rank per GPU")
    dev = torch.device("cuda", local_rank % max(n_dev, 1))  # ranks share a GPU only in gloo rehearsals
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if args.mode == "engine":
        if world > 1:
            dist.barrier()
        st, (eB, eS, eD, eV, e_blocks) = run_engine_mode(args, rank, world, dev)
        ref_shape = (eB, eS, eD) == (1024, 128, 2048)
        tok = torch.tensor([float(st.total_tokens)], device=dev)
        sec = torch.tensor([st.seconds], device=dev)
        if world > 1:
            dist.all_reduce(tok, op=dist.ReduceOp.SUM)
            dist.all_reduce(sec, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({
                "metric": "decode tokens/sec (whole node) on synthetic batch", "value": tok.item() / sec.item(),
                "unit": "tokens/s", "n_gpus": world, "steps": int(st.iterations), "warmup": 0,
                "ms_per_step": sec.item() / max(int(st.iterations), 1) * 1e3, "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": tok.item() / sec.item() / world / 123284.0 if ref_shape and args.engine_kind in ("paged", "paged_gemm") else None,
                "dtype": {"paged_bf16": "bf16", "paged_fp8": "fp8"}.get(args.engine_kind, "f32"),
                "data": "synthetic",
                "config": {"workload": ("engine: reference tests/paged_for_profile.cpp workload" if ref_shape else
                                        f"engine: {args.workload} shape under the reference's profiling recipe") +
                                       f" (B={eB} slots, S={eS}, D={eD}, V={eV}, {e_blocks} pages, {2 * eB} items, "
                                       f"prompt U[1,64]), {args.engine_kind} engine x{args.engine_replicas}, {'sequential' if args.sequential_loop or args.reference_quirk else 'pipelined'} loop, ThroughputCounter tokens/s incl. "
                                       "host scheduling, prefill, page growth and preemption",
                           "vs_baseline_note": "per-GPU value / README.md:79-82 (123284 tok/s, unnamed NVIDIA GPU)",
                           "reference_length_reset_quirk": bool(args.reference_quirk),
                           "layers": "reference launch sequence" if args.reference_launch_sequence else "lean compositions",
                           "repeat": st.repeat, "total_tokens": tok.item(), "seconds": sec.item()}}))
        if world > 1:
            dist.destroy_process_group()
        return

    side = torch.cuda.Stream(device=dev)  # a real stream: the legacy default stream cannot be captured into a graph
    with torch.cuda.stream(side):
        out = run_step_bench(args, rank, world, dev, dist)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # ranks leave together (rank 0 may still have been in its roofline pass)
        dist.destroy_process_group()


def timed_steps(wl, step, steps, warmup, world, dist, gather=None, repeats=1):
    """W untimed steps, then `repeats` timed regions of EXACTLY K steps each, every region bracketed by barrier +
    synchronize on both sides and started from the same state (the lengths are put back outside the timed region).
    Returns the list of per-region wall times of this rank (the caller takes the max over ranks per region)."""
    def one():
        if gather is not None:
            # the decoder writes this step's tokens into a buffer whose previous gather has completed; the gather of
            # this step (the path's only exchange: 4 KiB of token ids per rank at B=1024) then runs on RCCL's stream
            # beside the next step's kernels
            wl.decoder_result = gather.buffer().view(wl.B, 1)
        step()
        if gather is not None:
            gather()

    for _ in range(warmup):
        one()
    if gather is not None:
        gather.wait()
    regions = []
    for _ in range(max(1, repeats)):
        wl.lengths.copy_(wl.lengths0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        if gather is not None:
            gather.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        regions.append(time.perf_counter() - t0)
        # every row must still be live, otherwise "tokens" would over-count
        grown = (wl.lengths - wl.lengths0).cpu().numpy()
        assert (grown == steps).all(), "a row finished during the timed region"
    return regions


def region_stats(regions, steps):
    """ms per step of the repeated K-step regions: the median region is the headline, min / max say how far to trust it."""
    ms = sorted(r / steps * 1e3 for r in regions)
    return {"regions": len(ms), "median_ms_per_step": ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]),
            "min_ms_per_step": ms[0], "max_ms_per_step": ms[-1], "all_ms_per_step": [r / steps * 1e3 for r in regions]}


def median_region(regions):
    r = sorted(regions)
    return r[len(r) // 2] if len(r) % 2 else 0.5 * (r[len(r) // 2 - 1] + r[len(r) // 2])


def roofline_report(wl, workload, dtype, lengths_now, ms_per_step, reps, lean=True):
    """Per-launch times of one decode step (HIP events on the launch stream; launches of a few microseconds are
    replayed from a graph of 10 so that Python's launch rate is not what gets timed) and the roofline of the
    dominant kernel: its ALGORITHMIC bytes / its average launch duration against the 8 TB/s HBM peak."""
    alg = wl.algorithmic_bytes(lengths_now)
    L0 = wl.lengths.clone()
    times = {}
    for name, fn in wl.kernels(lean=lean).items():
        big = name.startswith("fused_decode_scan") and wl.B * wl.S >= (1 << 21)
        times[name] = time_kernel(fn, max(10, reps), batch=1 if big else 10)
        wl.lengths.copy_(L0)  # the decoder head advances the lengths
    torch.cuda.synchronize()
    if wl.layout == "paged":
        key_of = {"scan_lean" if lean else "scan": SCAN_LEAN if lean else SCAN_FULL}
    elif lean:
        key_of = {"scan_lean": SCAN_NAIVE}
    else:
        key_of = {"qkt": "qkt", "softmax_v": "softmax_v"}
    dom = max(key_of, key=lambda k: times[key_of[k]])
    ms = times[key_of[dom]]
    achieved = alg[dom] / (ms * 1e-3) / 1e9
    traffic, traffic_src, traffic_run_bytes = pmc_traffic(workload, dom, wl.layout, dtype)
    return {
        "bound": "hbm", "kernel": key_of[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
        # the counter passes ran fewer steps than this run (shorter rows): their own algorithmic bytes, so that the ratio
        # compares like with like (1.0 = no re-reads)
        "traffic_over_algorithmic_of_counter_run": (traffic / traffic_run_bytes) if traffic and traffic_run_bytes else None,
        "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": ms, "kernel_ms": times,
        "sum_of_launches_ms": sum(times.values()),
        "step_algorithmic_bytes": alg["step"],
        "step_gbs": alg["step"] / (ms_per_step * 1e-3) / 1e9,
        "step_frac": alg["step"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }, times


def side_config(name, dtype, dev, steps, warmup, repeats, cpu_budget_s=0.0):
    """One of the other BASELINE configurations as a short single-GPU step run (the `configs` block of the line)."""
    cfg_index = sorted(WORKLOADS).index(name) + 1
    wl = Workload(name, dev, 0x5EED0000 + cfg_index * 16, headroom=steps + warmup + 8, dtype=dtype)
    wl.lean_step()
    wl.lengths.copy_(wl.lengths0)
    regions = timed_steps(wl, wl.lean_step, steps, warmup, 1, None, repeats=repeats)
    elapsed = median_region(regions)
    ms = elapsed / steps * 1e3
    roof, _ = roofline_report(wl, name, dtype, wl.lengths.cpu().numpy(), ms, max(20, steps // 2))
    r = {"workload": f"{name}: {wl.layout} KV decode step, {wl.B} rows, emb_dim {wl.D}, max_seq {wl.S}, "
                     f"{ELEM_NAMES[dtype]}, lean composition",
         "dtype": dtype, "steps": steps, "warmup": warmup, "value": wl.B * steps / elapsed, "unit": "tokens/s",
         "ms_per_step": ms, "repeat": region_stats(regions, steps), "roofline": roof}
    if cpu_budget_s > 0:
        import types
        r["cpu_baseline"] = cpu_baseline(types.SimpleNamespace(B=wl.B, D=wl.D, S=wl.S, lengths_host=wl.lengths_host),
                                         budget_s=cpu_budget_s)
    del wl
    torch.cuda.empty_cache()
    return r


def engine_config(args, dev):
    """The reference's own profiling workload end to end (README.md:54-82: 123 284 tok/s on an unnamed NVIDIA GPU),
    in a child process so that the engine's allocations and host threads never share this process's timed region.
    Every configuration is run --engine-repeats times inside its child (a fresh engine each time, same items); `value`
    is the median run, min / max beside it -- one run is 0.15 s, too short to compare two compositions on."""
    import subprocess
    out = {}
    for label, extra in (("e1_f32_paged_gemm", []), ("e1_f32_paged_gemm_sequential_loop", ["--sequential-loop"]),
                         ("e1_f32_paged_gemm_sequential_loop_reference_launch_sequence",
                          ["--sequential-loop", "--reference-launch-sequence"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", "engine", "--gpus", "1", "--engine-repeats",
               str(args.engine_repeats), *extra]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
            d = json.loads(line)
            out[label] = {"value": d["value"], "unit": d["unit"], "vs_readme_123284": d["vs_baseline"],
                          "iterations": d["steps"], "seconds": d["config"]["seconds"], "repeat": d["config"].get("repeat"),
                          "workload": d["config"]["workload"]}
        except Exception as e:  # a side measurement never takes the bench line down
            out[label] = {"error": str(e)[:300]}
    return out


def collective_report(wl, step, args, world, dist, gather, dev, ms_with_gather):
    """What the N > 1 line carries so that the record proves the collective ran over N ranks: the backend, the number
    of ranks an all-reduce of ones saw, what the gathered token buffer holds, the all-gather's own duration, and the
    step time with the gather left out (same state, same K)."""
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)                                          # over the same communicator as the token gather
    ranks_seen = int(ones.item())
    assert ranks_seen == world, f"all-reduce saw {ranks_seen} ranks, WORLD_SIZE is {world}"
    gather.wait()
    latest = gather.latest()
    assert latest.numel() == world * wl.B
    n_valid = int((latest >= 0).sum().item())                     # every row of every rank is live: no EMPTY (-1) ids
    # the all-gather on its own: K blocking gathers of the ranks' [rows] int32 buffers
    buf, outb = gather.inputs[0], gather.outputs[0]
    for _ in range(3):
        dist.all_gather_into_tensor(outb, buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dist.all_gather_into_tensor(outb, buf)
    torch.cuda.synchronize()
    t_gather = torch.tensor([(time.perf_counter() - t0) / args.steps], device=dev)
    dist.all_reduce(t_gather, op=dist.ReduceOp.MAX)
    # the same K steps without the gather
    regions = timed_steps(wl, step, args.steps, args.warmup, world, dist, None, repeats=max(1, args.repeats))
    t = torch.tensor(regions, device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {"backend": dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else ""),
            "ranks_seen": ranks_seen, "gathered_token_ids": latest.numel(), "gathered_token_ids_valid": n_valid,
            "bytes_per_rank_per_step": wl.B * 4, "gather_us_per_step": t_gather.item() * 1e6,
            "ms_per_step_without_gather": median_region(t.tolist()) / args.steps * 1e3,
            "ms_per_step_with_gather": ms_with_gather,
            "note": "the gather is asynchronous (RCCL's stream, beside the next step's kernels), so its blocking "
                    "duration is not added to the step"}


def run_step_bench(args, rank, world, dev, dist):
    cfg_index = sorted(WORKLOADS).index(args.workload) + 1
    wl = Workload(args.workload, dev, 0x5EED0000 + cfg_index * 16 + rank, headroom=args.steps + args.warmup + 8,
                  dtype=args.dtype)
    from min_llm_inference_amd.sharding import TokenGather
    gather = TokenGather(wl.B, world, dev)
    lean = not args.materialising
    step = wl.lean_step if lean else wl.step
    step()                                  # per-stream workspaces are allocated outside the timed region
    wl.lengths.copy_(wl.lengths0)
    regions = timed_steps(wl, step, args.steps, args.warmup, world, dist, gather, repeats=max(1, args.repeats))
    tmax = torch.tensor(regions, device=dev, dtype=torch.float64)
    tokens = torch.tensor([float(wl.B * args.steps)], device=dev)
    if world > 1:
        dist.all_reduce(tokens, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)           # per region: the slowest rank
    regions = tmax.tolist()
    elapsed = median_region(regions)
    total_tokens = float(tokens.item())
    form = ("lean composition: what the layers run -- no scores / probabilities / emb_score materialised, chunks merged "
            "inside the scan launch" if lean else
            "materialising composition: the reference's launch sequence, qkt_output probabilities and emb_score written")
    name = args.workload
    if args.workload == "c4" and world == 8:
        name = "c5: 8192 rows over 8 GPUs = c4"          # BASELINE.json configs[4]
    out = {
        "metric": "decode tokens/sec (whole node) on synthetic batch",
        "value": total_tokens / elapsed,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16", "fp8": "fp8"}[args.dtype],
        "data": "synthetic",
        "repeat": dict(region_stats(regions, args.steps),
                       note="every region = exactly `steps` steps from the same state, barrier + synchronize on both sides, "
                            "max over ranks; `value` / `ms_per_step` are the median region"),
        "config": {
            "workload": f"{name}: {wl.layout} KV decode step (attention + greedy decoder head, n_new=0), "
                        f"{wl.B} rows/GPU, emb_dim {wl.D}, max_seq {wl.S}, lengths U[{wl.S // 4},{int(wl.lengths_host.max())}], "
                        f"{ELEM_NAMES[args.dtype]}{'' if args.dtype == 'f32' else ', fp32 accumulate'}; {form}",
            "rows_per_gpu": wl.B, "rows_total": wl.B * world, "emb_dim": wl.D, "max_seq": wl.S, "n_vocab": N_VOCAB,
            "mean_length": float(wl.lengths_host.mean()),
            "parallelism": f"row-sharded replicas x{world}, all-gather of token ids",
        },
    }
    if world > 1:
        coll = collective_report(wl, step, args, world, dist, gather, dev, out["ms_per_step"])
        if rank == 0:
            out["collective"] = coll
    if rank == 0 and not args.no_roofline:
        lengths_now = wl.lengths.cpu().numpy()
        reps = max(10, args.steps)
        roof, times = roofline_report(wl, args.workload, args.dtype, lengths_now, out["ms_per_step"], reps, lean=lean)
        roof["projection_gemm"] = gemm_report(wl, lengths_now, times)
        out["roofline"] = roof
        if world == 1 and wl.layout == "paged" and args.dtype != "fp8":
            # the other form of the same step, same state, same run: K steps each, plus its scan / combine launches
            other = wl.step if lean else wl.lean_step
            other()
            wl.lengths.copy_(wl.lengths0)
            t_other = median_region(timed_steps(wl, other, args.steps, args.warmup, 1, None, repeats=max(1, args.repeats)))
            o_roof, _ = roofline_report(wl, args.workload, args.dtype, wl.lengths.cpu().numpy(),
                                        t_other / args.steps * 1e3, reps, lean=not lean)
            mine = {"ms_per_step": out["ms_per_step"], "value": out["value"], "scan_frac": roof["frac"],
                    "scan_ms": roof["avg_launch_ms"]}
            theirs = {"ms_per_step": t_other / args.steps * 1e3, "value": wl.B * args.steps / t_other,
                      "scan_frac": o_roof["frac"], "scan_ms": o_roof["avg_launch_ms"], "kernel_ms": o_roof["kernel_ms"]}
            out["modes"] = {"lean": mine if lean else theirs, "materialising": theirs if lean else mine,
                            "note": "tokens, lengths and pages / caches are identical between the two forms; attention_result "
                                    "is bit-identical where the lean scan runs the chunked grid and equal up to the fp32 rounding "
                                    "of the merge (<= 2e-5) where it runs in equal page shares (config 4) or as the contiguous "
                                    "single-launch scan (tests/test_lean_path_gpu.py, test_full_size_properties_gpu.py); `value` "
                                    "is the form config.workload names"}
        if world == 1:
            roof["projection_gemm_d2048"] = large_gemm_report(dev)
            roof["measured_read_gbs"] = measure_read_gbs(dev)
            roof["measured_read_note"] = ("a pure streaming read of 4 GiB (16-byte lane loads, non-temporal, 16 KiB per wave in "
                                          "flight) on this box: what the scan could reach with no arithmetic and no page table")
    import types
    meta = types.SimpleNamespace(B=wl.B, D=wl.D, S=wl.S, lengths_host=wl.lengths_host)  # all cpu_baseline needs
    del wl
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_configs:
        cfgs = {}
        others = [("c2", "f32"), ("c3", "f32"), ("c4", "f32"), ("c4", "bf16"), ("c4", "fp8")]
        for name, dtype in others:
            if name == args.workload and dtype == args.dtype:
                continue
            if dtype == "fp8" and not ops.has_fp8():
                continue
            try:
                cfgs[f"{name}_{dtype}"] = side_config(name, dtype, dev, args.config_steps, 10, max(1, args.repeats),
                                                      cpu_budget_s=0.0 if args.no_cpu_baseline or name == "c4" else 4.0)
            except Exception as e:
                cfgs[f"{name}_{dtype}"] = {"error": str(e)[:300]}
        cfgs["engine"] = engine_config(args, dev)
        out["configs"] = cfgs
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(meta)
    if rank == 0:
        try:   # nothing this process started may outlive it (engine children, thread pools)
            import psutil
            kids = psutil.Process().children(recursive=True)
            out["children_alive_at_exit"] = [" ".join(k.cmdline())[:80] for k in kids if k.is_running() and k.status() != psutil.STATUS_ZOMBIE]
        except Exception:
            pass
    return out


if __name__ == "__main__":
    main()
