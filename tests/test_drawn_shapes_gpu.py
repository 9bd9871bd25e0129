"""Randomised differential parity in the reference's own style: its tests draw B, S, the dims, the lengths and the set of
new rows afresh on every run (tests/test_utils.cpp:256-291, 293-350: B in [1, 100], S in 4 * [100, 200], dims in [100, 257] on
purpose not multiples of 16; paged: tests/test_utils.cpp:695-773, S in 16 * [4, 16], D a multiple of 4, shuffled page pool).
The other GPU test files pin fixed shape lists; here the shapes are DRAWN -- from a master seed, so a failure reproduces:

    MLI_DRAWN_SEED=<int> python -m pytest tests/test_drawn_shapes_gpu.py -m gpu      # a fresh draw, like one run of the reference's suite

Every draw runs the materialising composition (whole tensors against the CPU oracle, untouched regions included) and the lean
composition the layers run (attention_result, K / V state).  Batch sizes are drawn from the lower part of the reference's range
so that the single-threaded oracle finishes the file in about a minute."""
import os

import numpy as np
import pytest

from gpu_util import host, to_dev
from helpers import assert_close, naive_case, paged_case, well_posed_rows

pytestmark = pytest.mark.gpu
MASTER = int(os.environ.get("MLI_DRAWN_SEED", "20261005"))
N_DRAWS = 16


def _draws(kind):
    rng = np.random.default_rng([MASTER, 0 if kind == "naive" else 1])
    out = []
    for i in range(N_DRAWS):
        if kind == "naive":
            din, dout = int(rng.integers(100, 258)), int(rng.integers(100, 258))
            if i % 2:   # every second draw: an output dim the single-launch lean scan covers (a multiple of 4)
                dout = 4 * (dout // 4)
            out.append((int(rng.integers(1, 2 ** 31)), int(rng.integers(1, 65)), 4 * int(rng.integers(100, 201)), din, dout,
                        [None, 5, 3][i % 3], bool(i % 2)))
        else:
            out.append((int(rng.integers(1, 2 ** 31)), int(rng.integers(8, 161)), 16 * int(rng.integers(4, 17)),
                        4 * int(rng.integers(16, 129)), [None, 5, 3][i % 3], bool(i % 2)))
    return out


def _attention_close(oracle, c, got, conditioned, what):
    """1e-3 absolute (the reference's threshold) -- on the reference's own U(0, 1] data for the rows where that is a
    well-posed demand.  Scores there are ~1e4: a 200-term fp32 dot product of that size carries an error of ~0.03 whatever the
    order of its sum, and in a row whose two best scores are closer than helpers.well_posed_rows' gap that error moves the
    probabilities by p (1 - p) per unit of score and the output by up to that times the spread of V -- several 1e-2 were
    observed between this GPU and the CPU oracle, neither of them wrong.  Such near-tie rows keep what is certain: the output is
    a convex combination of the row's live V rows (finite, inside their per-column range); they must not be the majority."""
    if conditioned:
        return assert_close(got, c["attention_result"], what=what)
    raw = np.zeros_like(c["qkt_output"])
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], raw)
    ok = well_posed_rows(raw, c["lengths"])
    assert ok.sum() >= max(1, len(ok) // 4), "too few well-posed rows in this draw"
    assert_close(got[ok], c["attention_result"][ok], what=what + ", well-posed rows")
    assert np.isfinite(got).all(), what
    for b in np.nonzero(~ok)[0]:
        v = c["v_cache"][b, :int(c["lengths"][b])]
        assert (got[b] >= v.min(axis=0) - 1e-3).all() and (got[b] <= v.max(axis=0) + 1e-3).all(), f"{what}: row {b} leaves V's range"


@pytest.mark.parametrize("seed,B,S,Din,Dout,zero_every,conditioned", _draws("naive"))
def test_drawn_contiguous_composition(oracle, mli, dev, seed, B, S, Din, Dout, zero_every, conditioned):
    """inference_self_attention on a drawn shape (reference InferenceOptimizedSelfAttentionTest, ..._test.cpp:139-190)."""
    from min_llm_inference_amd import ops
    c = naive_case(seed, B, S, Din, Dout, conditioned=conditioned, zero_every=zero_every)
    d = to_dev(c, dev)
    lean = to_dev(c, dev)
    ops.inference_self_attention(d["inp"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["kt_cache"],
                                 d["v_cache"], d["q_output"], d["qkt_output"], d["attention_result"], c["n_new"])
    covered = Dout % 4 == 0   # the scan reads kt_cache [B, Dout, S] and v_cache [B, S, Dout] in 16-byte pieces (S is a multiple of 4)
    lean_args = (lean["inp"], lean["lengths"], lean["wk"], lean["wq"], lean["wv"], lean["new_batch_idx"], lean["kt_cache"],
                 lean["v_cache"], lean["q_output"], lean["attention_result"], c["n_new"])
    if covered:
        ops.self_attention_lean(*lean_args)
    else:   # the documented contract (mli_kernels.h): MLI_ERR_BAD_ARG, the layer then takes mli_inference_self_attention
        from min_llm_inference_amd import MliError
        with pytest.raises(MliError):
            ops.self_attention_lean(*lean_args)
    oracle.self_attention_inference_host(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"], c["kt_cache"],
                                         c["v_cache"], c["q_output"], c["qkt_output"], c["attention_result"], c["n_new"])
    for name in ("q_output", "kt_cache", "v_cache"):
        assert_close(host(d[name]), c[name], what=f"{name} (materialising)")
    _attention_close(oracle, c, host(d["attention_result"]), conditioned, "attention_result (materialising)")
    if covered:
        for name in ("kt_cache", "v_cache"):
            assert_close(host(lean[name]), c[name], what=f"{name} (lean)")
        _attention_close(oracle, c, host(lean["attention_result"]), conditioned, "attention_result (lean)")
    probs = host(d["qkt_output"])
    live = c["lengths"] > 0
    assert np.isfinite(probs).all() and np.allclose(probs[live].sum(axis=1), 1.0, atol=1e-4)
    for b in range(B):   # the zero tail (and empty rows: all zero)
        assert (probs[b, int(c["lengths"][b]):] == 0).all()
    if conditioned:
        assert_close(probs, c["qkt_output"], what="qkt_output (probabilities)")


@pytest.mark.parametrize("seed,B,S,D,zero_every,conditioned", _draws("paged"))
def test_drawn_paged_composition(oracle, mli, dev, seed, B, S, D, zero_every, conditioned):
    """paged_attention on a drawn shape, compared through the page layout as the reference's tier-2 tests do
    (paged_attention_kernels_test.cpp:114-233: assert_page_table_close for K and V, q_output, attention_result)."""
    import torch
    from min_llm_inference_amd import ops
    c = paged_case(seed, B, S, D, conditioned=conditioned, zero_every=zero_every)
    oracle.clone_to_pages(c["pool"], c["table"], c["inp_embedding"], c["kt_cache"], c["v_cache"], c["lengths"])
    d = to_dev(c, dev)
    lean = to_dev(c, dev)
    ops.paged_attention(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["q_output"],
                        d["qkt_output"], d["attention_result"], c["n_new"], S)
    ops.paged_attention_lean(lean["page_table"], lean["lengths"], lean["wk"], lean["wq"], lean["wv"], lean["new_batch_idx"],
                             lean["q_output"], lean["attention_result"], c["n_new"], S)
    torch.cuda.synchronize()
    oracle.self_attention_inference_host(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"], c["attention_result"],
                                         c["n_new"])
    assert_close(host(d["q_output"]), c["q_output"], what="q_output")
    _attention_close(oracle, c, host(d["attention_result"]), conditioned, "attention_result (materialising)")
    _attention_close(oracle, c, host(lean["attention_result"]), conditioned, "attention_result (lean)")
    new_rows = set(c["new_batch_idx"][:c["n_new"]].tolist())
    for which in (d, lean):
        pool = host(which["pool"])
        k_got = oracle.gather_from_pages(pool, c["table"], c["lengths"], S, D, 1)
        v_got = oracle.gather_from_pages(pool, c["table"], c["lengths"], S, D, 2)
        for b in range(B):
            L = int(c["lengths"][b])
            if L:
                rows = list(range(L)) if b in new_rows else [L - 1]
                assert_close(k_got[b, rows, :], c["kt_cache"][b][:, rows].T, what=f"K row {b}")
                assert_close(v_got[b, rows, :], c["v_cache"][b, rows, :], what=f"V row {b}")
    assert torch.equal(d["pool"], lean["pool"]), "pages: lean == materialising, bit for bit"


def _draws_bf16():
    rng = np.random.default_rng([MASTER, 2])
    return [(int(rng.integers(1, 2 ** 31)), int(rng.integers(8, 97)), 16 * int(rng.integers(4, 65)), 8 * int(rng.integers(8, 129)),
             [None, 5, 3][i % 3]) for i in range(N_DRAWS // 2)]


@pytest.mark.parametrize("seed,B,S,D,zero_every", _draws_bf16())
def test_drawn_paged_bf16_lean_composition(oracle, mli, dev, seed, B, S, D, zero_every):
    """The lean paged composition over bf16 pages (what PagedAttentionBf16Layer runs; BASELINE config 4's element type) on a drawn
    shape.  Unpinned by the reference (fp32 only): the expectation is the fp32 oracle on bf16-rounded inputs with K / V rounded
    where the kernels store them -- tests/test_paged_bf16_gpu.py's rule and tolerances (q 1e-4, attention_result 1e-3)."""
    from min_llm_inference_amd import ops
    from helpers import bf16_round
    from test_paged_bf16_gpu import _case
    c, d = _case(oracle, dev, seed, B, S, D, zero_every)
    ops.paged_attention_lean(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["q_output"],
                             d["attention_result"], c["n_new"], S, elem=ops.ELEM_BF16)
    oracle.fill_new_kt_v_cache(c["inp_embedding"], c["new_batch_idx"], c["lengths"], c["wk"], c["wv"], c["kt_cache"],
                               c["v_cache"], c["n_new"])
    oracle.get_latest_kt_q_v(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    c["kt_cache"], c["v_cache"] = bf16_round(c["kt_cache"]), bf16_round(c["v_cache"])
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    live = c["lengths"] > 0
    assert_close(host(d["q_output"])[live], c["q_output"][live], thr=1e-4, what="q_output")
    assert_close(host(d["attention_result"]), c["attention_result"], what="attention_result")
