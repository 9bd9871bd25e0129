"""CPU tests of the oracle itself (oracle/oracle_cpu.c).

The reference holds no golden vectors for this path and cannot be built here (see oracle_cpu.c header:
PARITY UNPINNED), so the restatement is checked three ways: against an independent float64 numpy model
of the same mathematics, against domain properties the reference's tests rely on (tail zeros, untouched
regions, empty rows), and against the committed regression vectors in tests/golden/.
"""
import os

import numpy as np
import pytest

from helpers import assert_close, assert_equal, naive_case, paged_case

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_f64(c):
    """Independent float64 model of self_attention_inference_host (tests/test_utils.cpp:502-519)."""
    inp = c["inp"].astype(np.float64)
    wk, wq, wv = (c[k].astype(np.float64) for k in ("wk", "wq", "wv"))
    kt = c["kt_cache"].astype(np.float64).copy()
    v = c["v_cache"].astype(np.float64).copy()
    q = c["q_output"].astype(np.float64).copy()
    qkt = c["qkt_output"].astype(np.float64).copy()
    att = c["attention_result"].astype(np.float64).copy()
    B, S, _ = inp.shape
    D = wk.shape[1]
    for b in c["new_batch_idx"][:c["n_new"]]:
        L = int(c["lengths"][b])
        kt[b, :, :L] = (inp[b, :L] @ wk).T
        v[b, :L] = inp[b, :L] @ wv
    for b in range(B):
        L = int(c["lengths"][b])
        if L == 0:
            qkt[b] = 0
            att[b] = 0
            continue
        x = inp[b, L - 1]
        kt[b, :, L - 1] = x @ wk
        v[b, L - 1] = x @ wv
        q[b] = x @ wq
        s = (q[b] @ kt[b, :, :L]) / np.sqrt(np.float32(D))
        p = np.exp(s - s.max())
        p /= p.sum()
        qkt[b, :L] = p
        qkt[b, L:] = 0
        att[b] = p @ v[b, :L]
    return kt, v, q, qkt, att


@pytest.mark.parametrize("seed,B,S,Din,Dout", [(1, 4, 128, 64, 64), (2, 7, 104, 101, 57), (3, 3, 256, 32, 100)])
def test_composition_matches_float64_model(oracle, seed, B, S, Din, Dout):
    c = naive_case(seed, B, S, Din, Dout, conditioned=True, zero_every=3)
    kt, v, q, qkt, att = model_f64(c)
    oracle.self_attention_inference_host(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["new_batch_idx"],
                                         c["kt_cache"], c["v_cache"], c["q_output"], c["qkt_output"],
                                         c["attention_result"], c["n_new"])
    assert_close(c["kt_cache"], kt, thr=1e-4, what="kt")
    assert_close(c["v_cache"], v, thr=1e-4, what="v")
    assert_close(c["q_output"], q, thr=1e-4, what="q")
    assert_close(c["qkt_output"], qkt, thr=1e-5, what="probs")
    assert_close(c["attention_result"], att, thr=1e-4, what="attention")


def test_each_op_touches_only_what_the_reference_touches(oracle):
    c = naive_case(5, 6, 64, 24, 40, lengths=[0, 1, 17, 63, 64, 30])
    before = {k: v.copy() for k, v in c.items() if isinstance(v, np.ndarray)}
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    for b, L in enumerate(c["lengths"]):
        assert_equal(c["qkt_output"][b, L:], before["qkt_output"][b, L:], "qkt tail untouched")
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    for b, L in enumerate(c["lengths"]):
        assert (c["qkt_output"][b, L:] == 0).all()
        if L:
            assert abs(float(c["qkt_output"][b, :L].astype(np.float64).sum()) - 1.0) < 1e-5
    oracle.get_latest_kt_q_v(c["inp"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    assert_equal(c["q_output"][0], before["q_output"][0], "empty row q untouched")
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    assert (c["attention_result"][0] == 0).all()


def test_page_layout_round_trip_and_rule(oracle):
    c = paged_case(9, 9, 64, 12, lengths=[0, 1, 15, 16, 17, 63, 33, 48, 2])
    pool0 = c["pool"].copy()
    oracle.clone_to_pages(c["pool"], c["table"], c["inp_embedding"], c["kt_cache"], c["v_cache"], c["lengths"])
    S, D = 64, 12
    for seg, src in ((0, c["inp_embedding"]), (2, c["v_cache"]), (1, np.ascontiguousarray(c["kt_cache"].transpose(0, 2, 1)))):
        got = oracle.gather_from_pages(c["pool"], c["table"], c["lengths"], S, D, seg)
        for b, L in enumerate(c["lengths"]):
            assert_equal(got[b, :L], src[b, :L], f"seg {seg} row {b}")
    # the rule itself, spelled out once: (s % 16) * 3 * D + seg * D + d inside block table[b, s / 16]
    b, s, d = 5, 37, 7
    off = c["table"][b, s // 16] + (s % 16) * 3 * D + 1 * D + d
    assert c["pool"][off] == c["kt_cache"][b, d, s]
    # one position past the length is cloned (decoder's next embedding), nothing beyond it
    b, L = 6, 33
    off = c["table"][b, L // 16] + (L % 16) * 3 * D
    assert_equal(c["pool"][off:off + D], c["inp_embedding"][b, L])
    off2 = c["table"][b, (L + 1) // 16] + ((L + 1) % 16) * 3 * D
    assert_equal(c["pool"][off2:off2 + D], pool0[off2:off2 + D])


def test_decoder_semantics(oracle):
    """tests/decoder_test.cpp:46-94 edge rows: L = S-1 finishes, L = 0 yields EMPTY_ROW_TOKEN_ID, EOF finishes."""
    rng = np.random.default_rng(4)
    B, V, S, D = 5, 1030, 32, 16
    emb = rng.random((V, D), dtype=np.float32)
    wpe = rng.random((S, D), dtype=np.float32)
    x = rng.random((B, D), dtype=np.float32)
    emb[oracle.EOF_TOKEN_ID] *= 0.0
    lengths = np.array([0, S - 1, 3, 7, 1], np.int32)
    inp = rng.random((B, S, D), dtype=np.float32)
    inp0 = inp.copy()
    score = np.zeros((B, V), np.float32)
    res = np.full((B,), 99, np.int32)
    oracle.decoder_host(x, emb, score, wpe, inp, lengths, res)
    exp_tok = (x.astype(np.float64) @ emb.astype(np.float64).T).argmax(1)
    assert res[0] == oracle.EMPTY_ROW_TOKEN_ID and lengths[0] == 0
    assert res[1] == exp_tok[1] and lengths[1] == 0            # L + 1 >= S: finished, no embedding written
    assert_equal(inp[1], inp0[1])
    assert res[2] == exp_tok[2] and lengths[2] == 4
    assert_close(inp[2, 3], emb[res[2]] + wpe[3], thr=0, what="next embedding")


@pytest.mark.parametrize("name", ["c1_naive", "odd_naive", "paged_small"])
def test_golden_vectors(oracle, name):
    """Regression vectors generated by tests/golden/make_golden.py from this oracle (NOT reference outputs)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    from golden.make_golden import run_case
    out = run_case(oracle, {k: g[k].copy() for k in g.files if k.startswith("in_")})
    for k in g.files:
        if k.startswith("out_"):
            assert_equal(out[k], g[k], f"{name}:{k}")


def test_fp8_helper_codes_match_the_table_search_model():
    """tests/helpers.fp8_bits (integer arithmetic on the float32 bits: what the fp8 GPU tests round their pools with) against
    the format's definition spelled out (nearest representable value by table search, ties to the even code, saturating at 448,
    NaN -> 0x7f): every code, every midpoint between two codes and its two float32 neighbours, the saturation edge, the subnormal
    grid, and random values of three scales -- both signs."""
    import helpers as h
    rng = np.random.default_rng(5)
    t = h._fp8_table()
    mids = ((t[:-1] + t[1:]) / 2).astype(np.float32)
    vals = np.concatenate([t, mids, np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9)),
                           [448, 449, 463.9, 464, 465, 1e9, 0, 1e-10, 2 ** -10, 2 ** -9, 2 ** -6, 2 ** -6 * 0.999],
                           rng.standard_normal(100000) * 3, rng.random(50000) * 600, (rng.random(50000) - 0.5) * 0.05]).astype(np.float32)
    vals = np.concatenate([vals, -vals, [np.nan]]).astype(np.float32)
    got, want = h.fp8_bits(vals), h.fp8_bits_by_search(vals)
    assert (got == want).all(), vals[got != want][:8]
    assert (h.fp8_decode(np.arange(127, dtype=np.uint8)).astype(np.float64) == t).all()
    assert np.isnan(h.fp8_decode(np.uint8(0x7f))) and h.fp8_decode(np.uint8(0xfe)) == -448
