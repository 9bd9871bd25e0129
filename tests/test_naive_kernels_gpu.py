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


# ---- the lean contiguous composition: one scan launch, no scores / probabilities in memory -----------------------
# (seed, B, S, Din, Dout): one chunk per row; several chunks, ragged; config 2 reduced; 512- and 1024-wide rows (two lane
# loads per V row / swept in slices); many short rows; S not a multiple of the chunk
LEAN_SHAPES = [(21, 33, 400, 257, 100), (22, 100, 800, 128, 128), (23, 4, 128, 64, 64), (24, 16, 1024, 256, 256),
               (25, 9, 2048, 64, 512), (26, 5, 516, 96, 1024), (27, 300, 260, 32, 36)]


@pytest.mark.parametrize("conditioned,zero_every", [(False, None), (True, 5)])
@pytest.mark.parametrize("seed,B,S,Din,Dout", LEAN_SHAPES)
def test_self_attention_lean(oracle, mli, dev, seed, B, S, Din, Dout, conditioned, zero_every):
    """mli_self_attention_lean (what SelfAttentionLayer runs) against the oracle's composition and against
    mli_inference_self_attention on the same inputs: same caches and q_output bit for bit, attention_result within the
    merge's fp32 rounding, qkt_output never touched."""
    from helpers import assert_equal
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout, conditioned=conditioned, zero_every=zero_every)
    lean, full = to_dev(c, dev), to_dev(c, dev)
    ops.self_attention_lean(lean["inp"], lean["lengths"], lean["wk"], lean["wq"], lean["wv"], lean["new_batch_idx"],
                            lean["kt_cache"], lean["v_cache"], lean["q_output"], lean["attention_result"], c["n_new"])
    ops.inference_self_attention(full["inp"], full["lengths"], full["wk"], full["wq"], full["wv"], full["new_batch_idx"],
                                 full["kt_cache"], full["v_cache"], full["q_output"], full["qkt_output"],
                                 full["attention_result"], c["n_new"])
    scratch_before = c["qkt_output"].copy()
    oracle.self_attention_inference_host(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"],
                                         c["attention_result"], c["n_new"])
    got = host(lean["attention_result"])
    if conditioned:
        assert_close(got, c["attention_result"], what="attention_result vs oracle")
    else:
        # reference distribution: scores of ~1e4 with an ulp of 1e-3 and a near-one-hot softmax -- a row whose two best
        # scores are within a few ulps turns one ulp of score rounding into 1e-3 of the result; compare where the
        # softmax is well-posed (helpers.well_posed_rows, as the probabilities of the materialising form are compared)
        from helpers import well_posed_rows
        raw = np.zeros_like(c["qkt_output"])
        oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], raw)
        ok = well_posed_rows(raw, c["lengths"]) | (c["lengths"] == 0)
        assert ok.sum() >= max(1, B // 4)
        assert_close(got[ok], c["attention_result"][ok], what="attention_result vs oracle (well-posed rows)")
    # the scores are computed exactly as launch_qkt computes them: the two forms differ by the merge's rounding only
    assert_close(got, host(full["attention_result"]), thr=2e-5, what="attention_result vs the materialising composition")
    for k in ("q_output", "kt_cache", "v_cache"):
        assert_equal(host(lean[k]), host(full[k]), what=k)
    assert_equal(host(lean["qkt_output"]), scratch_before, what="qkt_output must not be touched")


def test_self_attention_lean_chunk_edges(oracle, mli, dev):
    """Lengths on and around the 256-token chunk boundaries, the empty row, a single token, the full row."""
    from min_llm_inference_amd import ops
    L = [0, 1, 17, 255, 256, 257, 511, 512, 513, 768, 1023, 1024]
    c = naive_case(102, len(L), 1024, 64, 128, conditioned=True, lengths=L)
    c["n_new"] = 0
    d = to_dev(c, dev)
    ops.self_attention_lean(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["kt_cache"],
                            d["v_cache"], d["q_output"], d["attention_result"], 0)
    oracle.self_attention_inference_host(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"],
                                         c["attention_result"], 0)
    assert_close(host(d["attention_result"]), c["attention_result"], what="attention_result")
    # repeated launches: the rows' arrival counters are back at zero
    ops.self_attention_lean(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["kt_cache"],
                            d["v_cache"], d["q_output"], d["attention_result"], 0)
    assert_close(host(d["attention_result"]), c["attention_result"], what="attention_result, second launch")


def test_self_attention_lean_refuses_what_it_does_not_cover(mli, dev):
    """Dims that are not multiples of 4: MLI_ERR_BAD_ARG (the C++ adapter then takes inference_self_attention)."""
    from min_llm_inference_amd import ops
    c = naive_case(103, 7, 412, 101, 257)
    d = to_dev(c, dev)
    with pytest.raises(ops.MliError):
        ops.self_attention_lean(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["kt_cache"],
                                d["v_cache"], d["q_output"], d["attention_result"], c["n_new"])
