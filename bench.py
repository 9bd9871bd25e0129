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
import json
import os
import sys
import time

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


class Workload:
    """Synthetic decode state for one GPU (weights U(-1,1)/sqrt(D), lengths U[S/4, 3S/4],
    pages drawn from a shuffled pool)."""

    def __init__(self, name, dev, seed, headroom, dtype="f32"):
        self.name = name
        self.layout, self.B, self.D, self.S = WORKLOADS[name]
        self.dtype = dtype
        self.esize = 2 if dtype == "bf16" else 4  # bytes per KV / weight element
        if dtype == "bf16" and self.layout != "paged":
            raise SystemExit("bf16 is implemented for the paged layout (BASELINE config 4)")
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
        if dtype == "bf16":
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
                                    dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
            chunk = 1 << 28
            for o in range(0, self.pool.numel(), chunk):  # K/V/x contents: U(-1,1)
                n = min(chunk, self.pool.numel() - o)
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

    # ---- the individual kernels, for the roofline pass ------------------------------------
    def kernels_separate(self):
        return self.kernels(fused_scan=False)

    def kernels(self, fused_scan=True):
        w = self
        bf = self.dtype == "bf16"
        if fused_scan and self.layout == "paged" and self.D <= 2048:
            # the composition runs the single-pass scan (mli_decode_scan_paged): time its two launches apart
            latest = (ops.launch_get_latest_k_q_v_paged_attention_bf16 if bf else ops.launch_get_latest_k_q_v_paged_attention)
            return {
                "get_latest_k_q_v_paged (MFMA gather-GEMM-scatter)": lambda: latest(
                    w.page_table, w.lengths, w.wk, w.wq, w.wv, w.q_output, w.S),
                "fused_decode_scan (q.K^T + online softmax + softmax.V, one visit per page)": lambda: ops.decode_scan_paged(
                    w.q_output, w.page_table, w.lengths, w.qkt_output, w.attention_result, bf, phases=1),
                "fused_decode_combine": lambda: ops.decode_scan_paged(
                    w.q_output, w.page_table, w.lengths, w.qkt_output, w.attention_result, bf, phases=2),
            }
        if self.dtype == "bf16":
            return {
                "get_latest_k_q_v_paged_bf16 (MFMA gather-GEMM-scatter)": lambda: ops.launch_get_latest_k_q_v_paged_attention_bf16(
                    w.page_table, w.lengths, w.wk, w.wq, w.wv, w.q_output, w.S),
                "qkt_paged_bf16": lambda: ops.launch_qkt_paged_attention_bf16(w.q_output, w.page_table, w.lengths, w.qkt_output),
                "softmax_in_place_with_lengths": lambda: ops.launch_softmax_in_place_with_lengths(w.qkt_output, w.lengths),
                "softmax_v_paged_bf16": lambda: ops.launch_softmax_v_paged_attention_bf16(w.qkt_output, w.page_table,
                                                                                          w.attention_result, w.lengths),
            }
        if self.layout == "paged":
            return {
                "get_latest_k_q_v_paged (MFMA gather-GEMM-scatter)": lambda: ops.launch_get_latest_k_q_v_paged_attention(
                    w.page_table, w.lengths, w.wk, w.wq, w.wv, w.q_output, w.S),
                "qkt_paged": lambda: ops.launch_qkt_paged_attention(w.q_output, w.page_table, w.lengths, w.qkt_output),
                "softmax_in_place_with_lengths": lambda: ops.launch_softmax_in_place_with_lengths(w.qkt_output, w.lengths),
                "softmax_v_paged": lambda: ops.launch_softmax_v_paged_attention(w.qkt_output, w.page_table,
                                                                                w.attention_result, w.lengths),
            }
        return {
            "get_latest_kt_q_v (MFMA GEMM)": lambda: ops.launch_get_latest_kt_q_v(
                w.inp_embedding, w.lengths, w.wk, w.wq, w.wv, w.kt_cache, w.v_cache, w.q_output),
            "qkt": lambda: ops.launch_qkt(w.q_output, w.kt_cache, w.lengths, w.qkt_output),
            "softmax_in_place_with_lengths": lambda: ops.launch_softmax_in_place_with_lengths(w.qkt_output, w.lengths),
            "softmax_v": lambda: ops.launch_softmax_v(w.qkt_output, w.v_cache, w.attention_result, w.lengths),
        }

    def algorithmic_bytes(self, lengths):
        """Per-launch algorithmic HBM bytes (DESIGN.md 'Roofline accounting'); e = bytes per KV/weight element."""
        L = lengths.astype(np.int64)
        B, D, e = self.B, self.D, self.esize
        live = int((L > 0).sum())
        ptrs = int((8 * -(-L // PAGE)).sum()) if self.layout == "paged" else 0
        kv_one = int(L.sum()) * D * e
        qkt = kv_one + live * D * 4 + int(L.sum()) * 4 + ptrs + B * 4       # K + q in, scores out
        sv = kv_one + int(L.sum()) * 4 + live * D * 4 + ptrs + B * 4        # V + probs in, result out
        latest = live * D * (3 * e + 4) + 3 * D * D * e + B * 4 + (8 * live if self.layout == "paged" else 0)
        step = (2 * kv_one + live * (3 * D * e + D * 4) + 3 * D * D * e + B * 4 + ptrs)
        scan = 2 * kv_one + live * D * 4 + int(L.sum()) * 4 + ptrs + B * 4   # K + V + q in, raw scores out
        return {"qkt": qkt, "softmax_v": sv, "scan": scan, "get_latest": latest, "step": step}


def time_kernel(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps  # ms, on the stream the kernels were launched on


def measure_copy_gbs(dev):
    n = 1 << 28  # 1 GiB of floats in, 1 GiB out
    a = torch.empty(n, device=dev)
    b = torch.empty(n, device=dev)
    a.uniform_()
    ms = time_kernel(lambda: ops.stream_copy(a, b), 10)
    return 2 * n * 4 / (ms * 1e-3) / 1e9


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


MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


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
                      "118 TFLOP/s fp32 / 551 TFLOP/s bf16 (DESIGN.md 3.5)" if wl.D <= 512 else ""))}


def large_gemm_report(dev):
    """The same projection kernels where they are not latency-bound: x[1024, 2048].[Wk|Wq|Wv] (the reference's
    profiling shape, bench.py --workload e1), both element types, measured here so that the MFMA figure in the JSON
    is not only the 1.6-GFLOP launch of config 4."""
    out = {"shape": "B=1024, D=2048 (25.8 GFLOP per launch)", "unit": "TFLOP/s"}
    for dt in ("f32", "bf16"):
        try:
            w = Workload("e1", dev, 0xE1, headroom=8, dtype=dt)
            fn = [v for k, v in w.kernels().items() if k.startswith("get_latest")][0]
            ms = time_kernel(fn, 100)
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
        return None, None
    kernels = json.load(open(path))["kernels"]
    needle = {"qkt": "qkt_", "softmax_v": "softmax_v_partial", "scan": "fused_decode_scan"}[which]
    hits = [v["traffic_bytes"] for k, v in kernels.items() if needle in k]
    return (hits[0], os.path.relpath(path, ROOT)) if hits else (None, None)


def run_engine_mode(args, rank, world, dev):
    """Engine-level throughput on the reference's profiling workload (tests/paged_for_profile.cpp:11-23):
    B=1024 slots, S=128, D=2048, V=1024, 4096 pages, 2048 items with prompt length U[1,64], one forward round.
    tokens/s = ThroughputCounter (tokens appended / wall time, host scheduling and copies included) -- the
    figure README.md:54-82 quotes (123 284 tok/s on an unnamed NVIDIA GPU).  Each rank runs an independent
    replica of the workload on its own GPU."""
    from min_llm_inference_amd import engine as eng
    _, B, D, S = WORKLOADS[args.workload if args.workload != "c4" or args.engine_shape else "e1"]
    V = N_VOCAB
    rng = np.random.default_rng(0x5EED0100 + rank)

    def u(*shape, scale=1.0):
        return ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(scale)).astype(np.float32)

    emb = u(V, D)
    emb[ops.EOF_TOKEN_ID] *= 1.0001  # the reference scales the EOF row the same way
    kind = {"paged": eng.PAGED, "paged_gemm": eng.PAGED_GEMM, "paged_bf16": eng.PAGED_BF16}[args.engine_kind]
    # page pool: the reference gives 4 pages per slot at S = 128, i.e. half of the worst case B * S / 16 -- rows
    # outgrow it, so page growth and preemption are part of the measured run (mode E)
    n_blocks = B * S // 32
    R = max(1, args.engine_replicas)
    assert B % R == 0
    weights = (emb, u(S, D), u(D, D, scale=1 / np.sqrt(D)), u(D, D, scale=1 / np.sqrt(D)), u(D, D, scale=1 / np.sqrt(D)))
    items = [(i, rng.integers(0, ops.EOF_TOKEN_ID, size=int(rng.integers(1, 65)))) for i in range(2 * B)]
    engines = []
    for r in range(R):  # R engines of B / R slots each share the GPU (private streams, one host thread each)
        e = eng.Engine(kind, B // R, S, D, V, *weights, n_blocks=n_blocks // R, n_forward_rounds=1, device=dev.index,
                       reference_length_reset_quirk=args.reference_quirk)
        if R > 1:
            e.use_private_stream()
        if args.pipelined:
            e.set_pipelined()
        for i, toks in items[r::R]:
            e.add_item(i, toks)
        engines.append(e)
    if R == 1:
        st = engines[0].run()
        assert st.finished == 2 * B
        return st, (B, S, D, V, n_blocks)
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

    class Total:  # whole-GPU totals: tokens of all engines over the wall time of the slowest
        total_tokens = sum(s.total_tokens for s in stats)
        seconds = max(max(s.seconds for s in stats), wall)
        iterations = max(s.iterations for s in stats)
    return Total, (B, S, D, V, n_blocks)


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
    ap.add_argument("--dtype", choices=["auto", "f32", "bf16"], default="auto",
                    help="KV page / weight element type (accumulation, q, scores and outputs are always fp32); "
                         "auto = what BASELINE.json names for the workload: bf16 for c4, fp32 for c2/c3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--mode", choices=["step", "engine"], default="step",
                    help="step: kernel-level decode step (default); engine: the reference's profiling workload end to end")
    ap.add_argument("--engine-kind", choices=["paged", "paged_gemm", "paged_bf16"], default="paged_gemm",
                    help="paged_bf16 = extension: bf16 pages and weights (BASELINE config 4 dtype)")
    ap.add_argument("--pipelined", action="store_true",
                    help="engine mode: the pipelined loop (host one step behind the GPU; per-slot device updates)")
    ap.add_argument("--engine-replicas", type=int, default=1,
                    help="engine mode: split the slots over this many engines on the same GPU (private streams, one "
                         "host thread each) so one engine's host bookkeeping overlaps the other's kernels")
    ap.add_argument("--engine-shape", action="store_true",
                    help="engine mode: run the --workload shape (e.g. c4) instead of the reference's profiling shape e1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing "
                    "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="no GPU work: start the ranks, rendezvous over gloo, print one line (tests the launch path on CPU)")
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
                "vs_baseline": tok.item() / sec.item() / world / 123284.0 if ref_shape and args.engine_kind != "paged_bf16" else None,
                "dtype": "bf16" if args.engine_kind == "paged_bf16" else "f32",
                "data": "synthetic",
                "config": {"workload": ("engine: reference tests/paged_for_profile.cpp workload" if ref_shape else
                                        f"engine: {args.workload} shape under the reference's profiling recipe") +
                                       f" (B={eB} slots, S={eS}, D={eD}, V={eV}, {e_blocks} pages, {2 * eB} items, "
                                       f"prompt U[1,64]), {args.engine_kind} engine x{args.engine_replicas}{' pipelined' if args.pipelined else ''}, ThroughputCounter tokens/s incl. "
                                       "host scheduling, prefill, page growth and preemption",
                           "vs_baseline_note": "per-GPU value / README.md:79-82 (123284 tok/s, unnamed NVIDIA GPU)",
                           "reference_length_reset_quirk": bool(args.reference_quirk),
                           "total_tokens": tok.item(), "seconds": sec.item()}}))
        if world > 1:
            dist.destroy_process_group()
        return

    cfg_index = sorted(WORKLOADS).index(args.workload) + 1
    wl = Workload(args.workload, dev, 0x5EED0000 + cfg_index * 16 + rank, headroom=args.steps + args.warmup + 8,
                  dtype=args.dtype)
    from min_llm_inference_amd.sharding import TokenGather
    gather = TokenGather(wl.B, world, dev)

    def step():
        # the decoder writes this step's tokens into a buffer whose previous gather has completed; the gather of
        # this step (the path's only exchange: 4 KiB of token ids per rank at B=1024) then runs on RCCL's stream
        # beside the next step's kernels
        wl.decoder_result = gather.buffer().view(wl.B, 1)
        wl.step()
        gather()

    for _ in range(args.warmup):
        step()
    gather.wait()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gather.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    # every row must still be live, otherwise "tokens" below would over-count
    grown = (wl.lengths - wl.lengths0).cpu().numpy()
    assert (grown == args.steps + args.warmup).all(), "a row finished during the timed region"
    tokens = torch.tensor([float(wl.B * args.steps)], device=dev)
    tmax = torch.tensor([elapsed], device=dev)
    if world > 1:
        dist.all_reduce(tokens, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    total_tokens = float(tokens.item())

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
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {wl.layout} KV decode step (attention + greedy decoder head, n_new=0), "
                        f"{wl.B} rows/GPU, emb_dim {wl.D}, max_seq {wl.S}, lengths U[{wl.S // 4},{int(wl.lengths_host.max())}], "
                        f"{'bf16 pages and weights, fp32 accumulate' if args.dtype == 'bf16' else 'fp32'}",
            "rows_per_gpu": wl.B, "emb_dim": wl.D, "max_seq": wl.S, "n_vocab": N_VOCAB,
            "mean_length": float(wl.lengths_host.mean()),
            "parallelism": f"row-sharded replicas x{world}, all-gather of token ids",
        },
    }

    if rank == 0 and not args.no_roofline:
        lengths_now = wl.lengths.cpu().numpy()
        alg = wl.algorithmic_bytes(lengths_now)
        times = {}
        for name, fn in wl.kernels().items():
            times[name] = time_kernel(fn, max(10, args.steps))
        wl.lengths.copy_(torch.from_numpy(lengths_now).to(dev))
        if any(k.startswith("fused_decode_scan") for k in times):
            key_of = {"scan": [k for k in times if k.startswith("fused_decode_scan")][0]}
        else:
            key_of = {"qkt": [k for k in times if k.startswith("qkt")][0],
                      "softmax_v": [k for k in times if k.startswith("softmax_v")][0]}
        dom = max(key_of, key=lambda k: times[key_of[k]])
        ms = times[key_of[dom]]
        achieved = alg[dom] / (ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.workload, dom, wl.layout, args.dtype)
        out["roofline"] = {
            "bound": "hbm", "kernel": key_of[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": ms,
            "kernel_ms": times,
            "projection_gemm": gemm_report(wl, lengths_now, times),
            "projection_gemm_d2048": large_gemm_report(dev),
            "step_algorithmic_bytes": alg["step"],
            "step_gbs": alg["step"] / (out["ms_per_step"] * 1e-3) / 1e9,
            "measured_copy_gbs": measure_copy_gbs(dev),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # ranks leave together (rank 0 may still have been in its roofline pass)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
