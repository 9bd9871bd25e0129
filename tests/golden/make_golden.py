"""Generates the regression vectors under tests/golden/ from the CPU oracle (oracle/oracle_cpu.c).

    python tests/golden/make_golden.py

These are NOT outputs of the reference: it ships no fixtures and cannot be built in this image
(oracle_cpu.c header, DESIGN.md "Oracle").  They pin the oracle against silent change and give the GPU
tests a checker-independent set of input/expected-output pairs.  Inputs are seeded; each file holds
in_* arrays (inputs and the initial contents of in/out buffers) and out_* arrays (buffers after
self_attention_inference_host / after the page clone + same composition for the paged case).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import naive_case, paged_case  # noqa: E402

OUT_KEYS = ("kt_cache", "v_cache", "q_output", "qkt_output", "attention_result")


def run_case(oracle, inputs):
    c = {k[3:]: v for k, v in inputs.items()}
    n_new = int(c["n_new"])
    inp = c["inp_embedding"] if "inp_embedding" in c else c["inp"]
    oracle.self_attention_inference_host(inp, c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"],
                                         c["attention_result"], n_new)
    return {"out_" + k: c[k] for k in OUT_KEYS}


def make(oracle, name, case):
    ins = {"in_" + k: (np.asarray(v) if not isinstance(v, np.ndarray) else v.copy()) for k, v in case.items()}
    outs = run_case(oracle, {k: v.copy() for k, v in ins.items()})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **ins, **outs)
    print(name, {k: v.shape for k, v in outs.items()})


def main():
    import oracle
    c1 = naive_case(1001, 4, 128, 64, 64, conditioned=True, lengths=[0, 1, 17, 127])
    c1["new_batch_idx"][:4] = [1, 2, 3, 0]
    c1["n_new"] = 4
    make(oracle, "c1_naive", c1)
    make(oracle, "odd_naive", naive_case(1002, 5, 104, 101, 57, conditioned=True, zero_every=4))
    pg = paged_case(1003, 6, 64, 64, conditioned=True, lengths=[0, 1, 16, 17, 63, 40])
    make(oracle, "paged_small", pg)


if __name__ == "__main__":
    main()
