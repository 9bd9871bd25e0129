#!/usr/bin/env python3
"""Times the forms of one decode step side by side on a side stream (HIP events): materialising vs lean attention,
eager vs hipGraph replay, in-kernel merge vs combine launch, chunk sizes.  A tuning aid; bench.py is the record."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from min_llm_inference_amd import load_library, ops  # noqa: E402


def timed(fn, reps, stream):
    fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--dtype", default="auto")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--chunks", default="0")
    a = ap.parse_args()
    dtype = a.dtype if a.dtype != "auto" else ("bf16" if a.workload == "c4" else "f32")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = load_library()
    side = torch.cuda.Stream(device=dev)
    out = {"workload": a.workload, "dtype": dtype}
    with torch.cuda.stream(side):
        for ct in [int(x) for x in a.chunks.split(",")]:
            lib.mli_tune(b"chunk_tokens", ct)
            wl = bench.Workload(a.workload, dev, 0x5EED, headroom=8, dtype=dtype)
            L0 = wl.lengths.clone()
            bf = dtype == "bf16"
            r = {}

            def reset():
                wl.lengths.copy_(L0)

            def stepped(fn):  # lengths advance by one per step: rewind between measurements
                reset()
                t = timed(fn, a.reps if wl.S > 2 * a.reps + 600 else 20, side)
                reset()
                return t

            r["step_materialising_eager"] = stepped(wl.step)
            r["step_lean_eager"] = stepped(wl.lean_step)
            wl.lean_step(); reset()
            g = ops.StepGraph(wl.lean_step)
            r["step_lean_graph"] = stepped(g.launch)
            g2 = ops.StepGraph(wl.step)
            r["step_materialising_graph"] = stepped(g2.launch)
            if wl.layout == "paged":
                scan = lambda ph, q=True: ops.decode_scan_paged(wl.q_output, wl.page_table, wl.lengths,
                                                                wl.qkt_output if q else None, wl.attention_result, bf,
                                                                phases=ph, n_sequence=wl.S)
                r["scan_materialising"] = timed(lambda: scan(1), a.reps, side)
                r["combine_materialising"] = timed(lambda: scan(2), a.reps, side)
                r["scan_lean_merge_in_kernel"] = timed(lambda: scan(7, False), a.reps, side)
                lib.mli_tune(b"scan_merge", 0)
                r["scan_lean_no_merge"] = timed(lambda: scan(5, False), a.reps, side)
                r["combine_lean"] = timed(lambda: scan(6, False), a.reps, side)
                lib.mli_tune(b"scan_merge", 1)
                alg = wl.algorithmic_bytes(wl.lengths_host)
                r["scan_lean_tbs"] = (alg["scan"] - int(wl.lengths_host.sum()) * 4) / r["scan_lean_merge_in_kernel"] / 1e6
                r["scan_materialising_tbs"] = alg["scan"] / r["scan_materialising"] / 1e6
            # two micro-batches (row halves) on two streams: one half's memory-bound scan beside the other half's
            # latency-bound GEMMs / decoder head
            halves = [wl.rows(0, wl.B // 2), wl.rows(wl.B // 2, wl.B)]
            streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            for h, st in zip(halves, streams):
                with torch.cuda.stream(st):
                    h.lean_step()
                st.synchronize()
            reset()

            def overlapped():
                ev = torch.cuda.Event()
                ev.record(side)
                for h, st in zip(halves, streams):
                    st.wait_event(ev)
                    with torch.cuda.stream(st):
                        h.lean_step()
                    done = torch.cuda.Event()
                    done.record(st)
                    side.wait_event(done)

            r["step_lean_two_micro_batches_eager"] = stepped(overlapped)

            def forked():  # the same fork / join as graph edges: one replay per step
                for h, st in zip(halves, streams):
                    ops.stream_wait_stream(st, side)
                    with torch.cuda.stream(st):
                        h.lean_step()
                    ops.stream_wait_stream(side, st)

            gf = ops.StepGraph(forked)
            r["step_lean_two_micro_batches_graph"] = stepped(gf.launch)
            quarters = [wl.rows(i * wl.B // 4, (i + 1) * wl.B // 4) for i in range(4)]
            streams4 = streams + [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            for h, st in zip(quarters, streams4):
                with torch.cuda.stream(st):
                    h.lean_step()
                st.synchronize()
            reset()

            def forked4():
                for h, st in zip(quarters, streams4):
                    ops.stream_wait_stream(st, side)
                    with torch.cuda.stream(st):
                        h.lean_step()
                    ops.stream_wait_stream(side, st)

            gf4 = ops.StepGraph(forked4)
            r["step_lean_four_micro_batches_graph"] = stepped(gf4.launch)
            gd = ops.StepGraph(wl.fused_decoder)
            r["decoder_fused_graph"] = stepped(gd.launch)
            gd2 = ops.StepGraph(wl.decoder)
            r["decoder_materialising_graph"] = stepped(gd2.launch)
            out[f"chunk_{ct}"] = {k: round(v, 2) for k, v in r.items()}
            del wl, g, g2, gd, gd2
            torch.cuda.empty_cache()
    lib.mli_tune(b"chunk_tokens", 0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
