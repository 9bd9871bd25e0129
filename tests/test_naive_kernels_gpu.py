"""GPU parity, contiguous KV layout: HIP kernels (through the C ABI) vs the CPU oracle.

Mirrors the reference's tests/self_attention_inference_optimized_test.cpp:6-190 -- each of the five
ops, the composition, and the every-5th-row-empty case -- on seeded inputs.  Whole output tensors
are compared, and device and host copies start from identical random contents, so "regions the op
must not touch stay untouched" is checked too.  Tolerance: 1e-3 absolute (the reference's), NaN fails.
"""
import numpy as np
import pytest

from gpu_util import host, to_dev
from helpers import assert_close, naive_case

pytestmark = pytest.mark.gpu

# (seed, B, S, Din, Dout): the reference draws B in [1,100], S in 4*[100,200], dims in [100,257]
SHAPES = [
    (11, 7, 412, 101, 257),
    (12, 33, 400, 257, 100),
    (13, 100, 800, 128, 128),
    (14, 1, 404, 200, 203),
    (15, 4, 128, 64, 64),      # BASELINE config 1 shape
    (16, 16, 1024, 256, 256),  # config 2 shape, reduced batch
]


@pytest.mark.parametrize("seed,B,S,Din,Dout", SHAPES)
def test_fill_new_kt_v_cache(oracle, mli, dev, seed, B, S, Din, Dout):
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout)
    d = to_dev(c, dev)
    ops.launch_fill_new_kt_v_cache(d["inp"], d["new_batch_idx"], d["lengths"], d["wk"], d["wv"], d["kt_cache"],
                                   d["v_cache"], c["n_new"])
    oracle.fill_new_kt_v_cache(c["inp"], c["new_batch_idx"], c["lengths"], c["wk"], c["wv"], c["kt_cache"],
                               c["v_cache"], c["n_new"])
    assert_close(host(d["kt_cache"]), c["kt_cache"], what="kt_cache")
    assert_close(host(d["v_cache"]), c["v_cache"], what="v_cache")


@pytest.mark.parametrize("seed,B,S,Din,Dout", SHAPES)
def test_get_latest_kt_q_v(oracle, mli, dev, seed, B, S, Din, Dout):
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout, zero_every=5)
    d = to_dev(c, dev)
    ops.launch_get_latest_kt_q_v(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["kt_cache"], d["v_cache"],
                                 d["q_output"])
    oracle.get_latest_kt_q_v(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    assert_close(host(d["kt_cache"]), c["kt_cache"], what="kt_cache")
    assert_close(host(d["v_cache"]), c["v_cache"], what="v_cache")
    assert_close(host(d["q_output"]), c["q_output"], what="q_output")  # empty rows must stay untouched


@pytest.mark.parametrize("seed,B,S,Din,Dout", SHAPES)
def test_qkt(oracle, mli, dev, seed, B, S, Din, Dout):
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout, zero_every=7)
    d = to_dev(c, dev)
    ops.launch_qkt(d["q_output"], d["kt_cache"], d["lengths"], d["qkt_output"])
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    assert_close(host(d["qkt_output"]), c["qkt_output"], what="qkt_output")  # s >= length untouched


@pytest.mark.parametrize("seed,B,S,Din,Dout", SHAPES)
def test_softmax_in_place_with_lengths(oracle, mli, dev, seed, B, S, Din, Dout):
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout, zero_every=3)
    c["qkt_output"] = (c["qkt_output"] * 20 - 10).astype(np.float32)  # spread the logits
    d = to_dev(c, dev)
    ops.launch_softmax_in_place_with_lengths(d["qkt_output"], d["lengths"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    got = host(d["qkt_output"])
    assert_close(got, c["qkt_output"], thr=1e-6, what="softmax")
    for b in range(B):  # tail is exactly zero, not merely small
        assert (got[b, c["lengths"][b]:] == 0).all()


@pytest.mark.parametrize("seed,B,S,Din,Dout", SHAPES)
def test_softmax_v(oracle, mli, dev, seed, B, S, Din, Dout):
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout, zero_every=4)
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    d = to_dev(c, dev)
    ops.launch_softmax_v(d["qkt_output"], d["v_cache"], d["attention_result"], d["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    assert_close(host(d["attention_result"]), c["attention_result"], what="attention_result")


@pytest.mark.parametrize("conditioned,zero_every,fused", [(False, None, 1), (True, 5, 1), (True, None, 0), (False, 5, 0)])
@pytest.mark.parametrize("seed,B,S,Din,Dout", SHAPES)
def test_inference_self_attention(oracle, mli, dev, seed, B, S, Din, Dout, zero_every, conditioned, fused):
    """reference tests InferenceOptimizedSelfAttentionTest / ...ZeroLengthTest (…_test.cpp:139-190);
    fused = 1 / 0 forces the softmax-fused / three-launch form of the composition."""
    from min_llm_inference_amd import ops
    assert mli.mli_tune(b"fused_softmax", fused) == 0
    c = naive_case(seed, B, S, Din, Dout, conditioned=conditioned, zero_every=zero_every)
    d = to_dev(c, dev)
    ops.inference_self_attention(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["kt_cache"],
                                 d["v_cache"], d["q_output"], d["qkt_output"], d["attention_result"], c["n_new"])
    mli.mli_tune(b"fused_softmax", -1)
    oracle.self_attention_inference_host(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"],
                                         c["attention_result"], c["n_new"])
    assert_close(host(d["attention_result"]), c["attention_result"], what="attention_result")
    assert_close(host(d["q_output"]), c["q_output"], what="q_output")
    assert_close(host(d["kt_cache"]), c["kt_cache"], what="kt_cache")
    assert_close(host(d["v_cache"]), c["v_cache"], what="v_cache")
    probs = host(d["qkt_output"])
    if conditioned:
        assert_close(probs, c["qkt_output"], what="qkt_output (probabilities)")
    else:  # reference distribution: near-one-hot softmax; compare the rows where that is well-posed (helpers.py)
        from helpers import well_posed_rows
        raw = np.zeros_like(c["qkt_output"])
        oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], raw)
        ok = well_posed_rows(raw, c["lengths"])
        assert ok.sum() >= max(1, B // 4)
        assert_close(probs[ok], c["qkt_output"][ok], what="qkt_output (probabilities, well-posed rows)")
        live = c["lengths"] > 0
        assert np.isfinite(probs).all() and np.allclose(probs[live].sum(axis=1), 1.0, atol=1e-4)


def test_config1_exact_lengths(oracle, mli, dev):
    """BASELINE config 1: B=4, D=64, S=128 with the edge lengths {0, 1, 17, 127}."""
    from min_llm_inference_amd import ops
    c = naive_case(101, 4, 128, 64, 64, conditioned=True, lengths=[0, 1, 17, 127])
    c["new_batch_idx"][:4] = [1, 2, 3, 0]
    c["n_new"] = 4
    d = to_dev(c, dev)
    ops.inference_self_attention(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["kt_cache"],
                                 d["v_cache"], d["q_output"], d["qkt_output"], d["attention_result"], 4)
    oracle.self_attention_inference_host(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"],
                                         c["attention_result"], 4)
    for k in ("attention_result", "q_output", "qkt_output", "kt_cache", "v_cache"):
        assert_close(host(d[k]), c[k], what=k)
    assert (host(d["attention_result"])[0] == 0).all()  # empty row -> zeros
