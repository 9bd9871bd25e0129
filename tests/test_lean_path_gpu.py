"""GPU parity of what the layers and engines actually run:

* the LEAN single-pass scan -- no raw scores, no probabilities, every row's chunks merged inside the scan launch by the
  workgroup that completes the row (an inter-workgroup hand-off: write-through stores, arrival counter, acquire) --
  against the oracle AND bit for bit against the materialising two-launch form on the same inputs;
* the decoder head with the argmax as the logits GEMM's epilogue (no emb_score) against the oracle's decoder and the
  materialising launcher: tokens, lengths and the written embeddings must be identical, ties included;
* a decode step replayed from a hipGraph against the same step launched eagerly.
"""
import numpy as np
import pytest
import torch

from gpu_util import host, to_dev
from helpers import (PAGE, assert_close, assert_equal, bf16_bits, bf16_round, build_page_pool, paged_case, rand_f,
                     rand_i)

pytestmark = pytest.mark.gpu

SENTINEL = 12345.0


def _prepare(oracle, dev, seed, B, S, D, **kw):
    c = paged_case(seed, B, S, D, **kw)
    oracle.clone_to_pages(c["pool"], c["table"], c["inp_embedding"], c["kt_cache"], c["v_cache"], c["lengths"])
    return c, to_dev(c, dev)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# (seed, B, S, D): single chunk, several chunks with ragged lengths, wide rows (the waves split the row), odd width
SCAN_SHAPES = [(131, 24, 256, 512), (132, 9, 1024, 256), (133, 6, 4096, 512), (134, 40, 64, 64), (135, 12, 128, 2048),
               (136, 7, 2048, 1024), (137, 5, 4096, 1540), (138, 300, 1024, 128), (139, 700, 128, 64), (140, 1100, 32, 512)]


@pytest.mark.parametrize("merge", [1, 0])
@pytest.mark.parametrize("seed,B,S,D", SCAN_SHAPES)
def test_lean_scan_matches_oracle_and_the_materialising_form(oracle, mli, dev, seed, B, S, D, merge):
    from min_llm_inference_amd import ops
    c, d = _prepare(oracle, dev, seed, B, S, D, conditioned=True, zero_every=4)
    # materialising form first (scan + combine launches): the reference's contract, checked elsewhere against the oracle
    ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], d["qkt_output"], d["attention_result"], False)
    full = host(d["attention_result"]).copy()
    d["attention_result"].fill_(SENTINEL)
    d["qkt_output"].fill_(SENTINEL)
    assert mli.mli_tune(b"scan_merge", merge) == 0
    try:
        for _ in range(3):   # the arrival counters must be back at zero after every launch
            d["attention_result"].fill_(SENTINEL)
            ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], None, d["attention_result"], False,
                                  phases=7, n_sequence=S)
            assert_equal(host(d["attention_result"]), full, what="lean == materialising, bit for bit")
    finally:
        mli.mli_tune(b"scan_merge", 1)
    assert (host(d["qkt_output"]) == SENTINEL).all(), "lean mode must not touch qkt_output"
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    assert_close(full, c["attention_result"], what="attention_result vs oracle")
    assert (full[c["lengths"] == 0] == 0).all()


# (seed, B, S, D, lengths): long rows shared by many workgroups, short rows (several per workgroup, more than one group),
# empty rows in between, a single row, exactly-full pages
STREAM_CASES = [
    (191, 48, 4096, 512, None),
    (192, 300, 1024, 128, None),
    (193, 2048, 1024, 64, "short"),
    (194, 1, 4096, 256, [4095]),
    (195, 9, 1024, 1024, [0, 1023, 0, 0, 16, 17, 512, 1, 1008]),
    (196, 64, 2048, 256, "short"),
]


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("seed,B,S,D,lengths", STREAM_CASES)
@pytest.mark.parametrize("dyn", [12, 0])
def test_lean_scan_equal_page_shares(oracle, mli, dev, seed, B, S, D, lengths, bf16, dyn):
    """attention_stream.hip forced onto small shapes (its size threshold lowered): every row's result within 1e-5 of the
    chunked form (other split points), within 1e-3 of the oracle, empty rows zero, and bit-identical from launch to launch
    (the shares depend on the lengths only; rows merged in token order by whichever workgroup arrives last)."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    if isinstance(lengths, str):
        lengths = rng.integers(0, 70, size=B).astype(np.int32)      # many rows per workgroup share
    c, d = _prepare(oracle, dev, seed, B, S, D, conditioned=True, lengths=lengths, zero_every=None if lengths is not None else 7)
    if bf16:
        if D % 8:
            pytest.skip("bf16 rows are multiples of 8")
        c["pool"] = bf16_round(c["pool"])
        p16 = _t(bf16_bits(c["pool"]).view(np.int16), dev).view(torch.bfloat16)
        d["pool16"] = p16
        d["page_table"] = _t(np.where(c["table"] >= 0, p16.data_ptr() + 2 * c["table"], 0).astype(np.int64), dev)
    else:
        pass
    q = d["q_output"]
    mli.mli_tune(b"scan_stream", 0)
    try:
        ops.decode_scan_paged(q, d["page_table"], d["lengths"], None, d["attention_result"], bf16, phases=7, n_sequence=S)
        chunked = host(d["attention_result"]).copy()
        mli.mli_tune(b"scan_stream", 1)
        assert mli.mli_tune(b"scan_stream_min_tokens", 0) == 0
        assert mli.mli_tune(b"scan_stream_dynamic_pct", dyn) == 0   # 0: equal static shares only; 12: + granules by ticket
        assert mli.mli_tune(b"scan_stream_granule", 16) == 0
        outs = []
        for _ in range(3):
            d["attention_result"].fill_(SENTINEL)
            ops.decode_scan_paged(q, d["page_table"], d["lengths"], None, d["attention_result"], bf16, phases=7, n_sequence=S)
            outs.append(host(d["attention_result"]).copy())
    finally:
        mli.mli_tune(b"scan_stream", 1)
        mli.mli_tune(b"scan_stream_min_tokens", 1 << 21)
        mli.mli_tune(b"scan_stream_dynamic_pct", 4)
        mli.mli_tune(b"scan_stream_granule", 64)
    assert_equal(outs[1], outs[0], what="second launch")
    assert_equal(outs[2], outs[0], what="third launch")
    assert_close(outs[0], chunked, thr=1e-5, what="equal shares vs chunked grid")
    assert (outs[0][c["lengths"] == 0] == 0).all()
    # the oracle on what the pages hold: for bf16 pages that is the bf16-rounded K / V (rounding is element-wise, so
    # rounding the contiguous caches equals rounding the pool they were cloned into); q, scores and sums are fp32
    kt = bf16_round(c["kt_cache"]) if bf16 else c["kt_cache"]
    v = bf16_round(c["v_cache"]) if bf16 else c["v_cache"]
    oracle.qkt_host(c["q_output"], kt, c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], v, c["attention_result"], c["lengths"])
    assert_close(outs[0], c["attention_result"], what="attention_result vs oracle")


@pytest.mark.parametrize("S,gran", [(1024, 16), (1024, 64), (4096, 16), (2048, 32)])
def test_lean_scan_equal_page_shares_with_the_largest_dynamic_part(oracle, mli, dev, S, gran):
    """scan_stream_dynamic_pct at its clamp (60 %): static shares shrink to 6 pages, so a row is cut into the most triples
    the split can produce -- at most W / min(6, granule) + 4, which launch_stream_decode holds against the W / 4 workspace
    slots per row before it launches (nothing enforced that bound).  Full rows (L = S - 1) are the worst case."""
    from min_llm_inference_amd import ops
    B, D = 40, 128
    rng = np.random.default_rng(180 + S + gran)
    lengths = rng.integers(S // 2, S, size=B).astype(np.int32)
    lengths[:6] = [S - 1, S - 1, S - 2, 0, 1, S - 16]
    c, d = _prepare(oracle, dev, 181, B, S, D, conditioned=True, lengths=lengths)
    mli.mli_tune(b"scan_stream", 0)
    try:
        ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], None, d["attention_result"], False, phases=7, n_sequence=S)
        chunked = host(d["attention_result"]).copy()
        mli.mli_tune(b"scan_stream", 1)
        assert mli.mli_tune(b"scan_stream_min_tokens", 0) == 0
        assert mli.mli_tune(b"scan_stream_dynamic_pct", 60) == 0
        assert mli.mli_tune(b"scan_stream_granule", gran) == 0
        outs = []
        for _ in range(3):
            d["attention_result"].fill_(SENTINEL)
            ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], None, d["attention_result"], False, phases=7,
                                  n_sequence=S)
            outs.append(host(d["attention_result"]).copy())
    finally:
        mli.mli_tune(b"scan_stream", 1)
        mli.mli_tune(b"scan_stream_min_tokens", 1 << 21)
        mli.mli_tune(b"scan_stream_dynamic_pct", 4)
        mli.mli_tune(b"scan_stream_granule", 64)
    assert_equal(outs[1], outs[0], what="second launch")
    assert_equal(outs[2], outs[0], what="third launch")
    assert_close(outs[0], chunked, thr=1e-5, what="equal shares (60 % dynamic) vs chunked grid")
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    assert_close(outs[0], c["attention_result"], what="attention_result vs oracle")
    assert (outs[0][3] == 0).all()


@pytest.mark.parametrize("chunk,tail", [(64, 0), (128, 0), (256, 0), (512, 64), (512, 128), (1024, 256), (256, 256)])
def test_lean_scan_many_chunks_per_row(oracle, mli, dev, chunk, tail):
    """Small chunks: up to 64 arrivals per row, rows of every chunk count side by side (uneven load)."""
    from min_llm_inference_amd import ops
    B, S, D = 96, 4096, 128
    rng = np.random.default_rng(140)
    lengths = rng.integers(0, S, size=B).astype(np.int32)
    lengths[:8] = [0, 1, chunk - 1, chunk, chunk + 1, S - 1, 2 * chunk, 17]
    c, d = _prepare(oracle, dev, 141, B, S, D, conditioned=True, lengths=lengths)
    assert mli.mli_tune(b"chunk_tokens", chunk) == 0
    assert mli.mli_tune(b"scan_tail_tokens", tail) == 0
    try:
        ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], d["qkt_output"], d["attention_result"], False)
        full = host(d["attention_result"]).copy()
        for _ in range(2):
            d["attention_result"].fill_(SENTINEL)
            ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], None, d["attention_result"], False,
                                  phases=7, n_sequence=S)
            assert_equal(host(d["attention_result"]), full, what=f"lean == materialising at {chunk}-token chunks")
    finally:
        mli.mli_tune(b"chunk_tokens", 0)
        mli.mli_tune(b"scan_tail_tokens", 0)
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    assert_close(full, c["attention_result"], what="attention_result vs oracle")
    probs = host(d["qkt_output"])       # the materialising form's probabilities under the same item layout
    assert_close(probs, c["qkt_output"], what="probabilities")


@pytest.mark.parametrize("seed,B,S,D", [(151, 128, 64, 512), (152, 24, 208, 1024), (153, 16, 1024, 256), (154, 3, 4096, 512)])
def test_paged_attention_lean_equals_paged_attention(oracle, mli, dev, seed, B, S, D):
    """The layers' composition (fill -> latest -> lean scan) leaves the same pages, q_output and attention_result as the
    reference-shaped one; it just never writes qkt_output."""
    from min_llm_inference_amd import ops
    c, d = _prepare(oracle, dev, seed, B, S, D, conditioned=True, zero_every=5)
    d2 = to_dev(c, dev)
    ops.paged_attention(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["q_output"],
                        d["qkt_output"], d["attention_result"], c["n_new"], S)
    ops.paged_attention_lean(d2["page_table"], d2["lengths"], d2["wk"], d2["wq"], d2["wv"], d2["new_batch_idx"],
                             d2["q_output"], d2["attention_result"], c["n_new"], S)
    assert_equal(host(d2["pool"]), host(d["pool"]), what="page pool")
    assert_equal(host(d2["q_output"]), host(d["q_output"]), what="q_output")
    assert_equal(host(d2["attention_result"]), host(d["attention_result"]), what="attention_result")
    assert_equal(host(d2["qkt_output"]), c["qkt_output"], what="qkt_output untouched")
    oracle.self_attention_inference_host(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"],
                                         c["new_batch_idx"], c["kt_cache"], c["v_cache"], c["q_output"],
                                         c["qkt_output"], c["attention_result"], c["n_new"])
    assert_close(host(d2["attention_result"]), c["attention_result"], what="attention_result vs oracle")


def test_paged_attention_lean_bf16_equals_paged_attention_bf16(mli, dev):
    from min_llm_inference_amd import ops
    B, S, D = 48, 1024, 512
    rng = np.random.default_rng(160)
    lengths = rng.integers(0, S - 1, size=B).astype(np.int32)
    lengths[:3] = [0, 1, S - 2]
    pool, table = build_page_pool(rng, lengths, S, D)
    pool = ((rng.random(pool.shape, dtype=np.float32) * 2 - 1)).astype(np.float32)

    def state():
        p = _t(bf16_bits(pool).view(np.int16), dev).view(torch.bfloat16)
        ptrs = _t(np.where(table >= 0, p.data_ptr() + 2 * table, 0).astype(np.int64), dev)
        return p, ptrs

    w = [_t(bf16_bits(((rng.random((D, D), dtype=np.float32) * 2 - 1) / np.sqrt(D)).astype(np.float32)).view(np.int16), dev)
         .view(torch.bfloat16) for _ in range(3)]
    L = _t(lengths, dev)
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    p1, t1 = state()
    p2, t2 = state()
    q1, q2 = torch.zeros(B, D, device=dev), torch.zeros(B, D, device=dev)
    o1, o2 = torch.zeros(B, D, device=dev), torch.zeros(B, D, device=dev)
    s1 = torch.zeros(B, S, device=dev)
    ops.paged_attention_bf16(t1, L, w[0], w[1], w[2], idx, q1, s1, o1, 0, S)
    ops.paged_attention_lean(t2, L, w[0], w[1], w[2], idx, q2, o2, 0, S)
    torch.cuda.synchronize()
    assert torch.equal(p1.view(torch.int16), p2.view(torch.int16)) and torch.equal(q1, q2) and torch.equal(o1, o2)
    assert torch.isfinite(o2).all() and (o2[0] == 0).all()


# ---- decoder head: argmax as the GEMM epilogue -------------------------------------------------------------------
def _decoder_inputs(rng, B, S, D, V):
    emb = (rng.random((V, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    wpe = rand_f(rng, (S, D))
    att = (rng.random((B, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    lengths = rand_i(rng, (B,), S - 2)
    lengths[:4] = [0, S - 1, S - 2, 1]            # empty slot, finishes on length, last writable position, first token
    att[5] = 1.0
    emb[1023, :] = 2.0                            # row 5 is pushed onto EOF
    # ties: identical embedding rows give identical logits; the LOWER index must win, within a 64-column tile
    # (70 vs 75), across tiles of one workgroup row (3 vs 700) and against the last, partial tile
    emb[75] = emb[70]
    emb[700] = emb[3]
    last = V - 1 if V - 1 != 1023 else V - 2     # (never the EOF row)
    emb[last] = emb[130]
    att[6] = emb[70] * 4
    att[7] = emb[3] * 4
    att[8] = emb[130] * 4
    return emb, wpe, att, lengths


@pytest.mark.parametrize("seed,B,S,D,V", [(161, 64, 128, 132, 1500), (162, 200, 64, 512, 1024), (163, 12, 4096, 64, 1025),
                                          (164, 1024, 64, 256, 1024)])
def test_decoder_fused_contiguous(oracle, mli, dev, seed, B, S, D, V):
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, att, lengths = _decoder_inputs(rng, B, S, D, V)
    inp_emb = rand_f(rng, (B, S, D))
    score = np.zeros((B, V), np.float32)
    res = np.full((B,), 77, np.int32)
    d_inp, d_len, d_res = _t(inp_emb, dev), _t(lengths, dev), _t(res, dev)
    ops.decoder_fused(_t(att, dev), _t(emb, dev), _t(wpe, dev), d_inp, d_len, d_res)
    # the materialising launcher on the same inputs
    m_inp, m_len, m_res, m_score = _t(inp_emb, dev), _t(lengths, dev), _t(res, dev), _t(score, dev)
    ops.launch_decoder(_t(att, dev), _t(emb, dev), m_score, _t(wpe, dev), m_inp, m_len, m_res)
    oracle.decoder_host(att, emb, score, wpe, inp_emb, lengths, res)
    assert_equal(host(d_res), host(m_res), what="tokens: fused == materialising")
    assert_equal(host(d_res), res, what="decoder_result vs oracle")
    assert_equal(host(d_len), lengths, what="lengths")
    assert_equal(host(d_inp), inp_emb, what="inp_embedding (next token rows; everything else untouched)")
    assert res[0] == -1 and lengths[0] == 0 and lengths[1] == 0 and res[5] == 1023 and lengths[5] == 0
    assert res[6] == 70 and res[7] == 3 and res[8] == 130, "ties go to the lowest index"


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("seed,B,S,D,V", [(165, 64, 128, 136, 1500), (166, 100, 256, 512, 1024), (167, 1024, 64, 512, 1024)])
def test_decoder_fused_paged_multi_rounds(oracle, mli, dev, seed, B, S, D, V, bf16):
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, att, lengths = _decoder_inputs(rng, B, S, D, V)
    pool, table = build_page_pool(rng, lengths, S, D)
    n_rounds, i_dec = 3, 1
    score = np.zeros((B, V), np.float32)
    res = np.full((B, n_rounds), 77, np.int32)

    def state():
        if bf16:
            p = _t(bf16_bits(pool).view(np.int16), dev).view(torch.bfloat16)
            ptrs = _t(np.where(table >= 0, p.data_ptr() + 2 * table, 0).astype(np.int64), dev)
        else:
            p = _t(pool, dev)
            ptrs = _t(np.where(table >= 0, p.data_ptr() + 4 * table, 0).astype(np.int64), dev)
        return p, ptrs, _t(lengths, dev), _t(res, dev)

    p1, t1, l1, r1 = state()
    p2, t2, l2, r2 = state()
    ops.paged_decoder_fused(_t(att, dev), _t(emb, dev), _t(wpe, dev), t1, l1, r1, i_dec, bf16)
    m_score = _t(score, dev)
    fn = ops.launch_paged_attention_decoder_multi_rounds_bf16 if bf16 else ops.launch_paged_attention_decoder_multi_rounds
    fn(_t(att, dev), _t(emb, dev), m_score, _t(wpe, dev), t2, l2, r2, i_dec)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2) and torch.equal(l1, l2), "tokens / lengths: fused == materialising"
    assert torch.equal(p1.view(torch.int16 if bf16 else torch.int32), p2.view(torch.int16 if bf16 else torch.int32))
    if not bf16:
        oracle.paged_decoder_host(att, emb, score, wpe, pool, table, lengths, res, i_dec)
        assert_equal(host(r1), res, what="decoder_result vs oracle")
        assert_equal(host(l1), lengths, what="lengths vs oracle")
        assert_equal(host(p1), pool, what="page pool (next embeddings in segment 0; everything else untouched)")
    got = host(r1)
    assert (got[:, [0, 2]] == 77).all() and got[6, i_dec] == 70 and got[7, i_dec] == 3 and got[8, i_dec] == 130


def test_decoder_fused_needs_its_scratch(mli, dev):
    from min_llm_inference_amd import _lib
    z = torch.zeros(4, 64, device=dev)
    i4 = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = mli.mli_decoder_fused(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), i4.data_ptr(), i4.data_ptr(),
                               4, 64, 64, 64, None, 0, None)
    assert rc == -12  # MLI_ERR_WORKSPACE


@pytest.mark.parametrize("seed,B,S,D", [(141, 1000, 128, 128), (142, 2048, 48, 64), (143, 513, 64, 1024)])
def test_rows_handed_out_longest_first_change_nothing(oracle, mli, dev, seed, B, S, D):
    """One-workgroup-per-row grids (short sequences, more rows than workgroup slots) take the rows longest first
    (mli_tune "scan_row_order"): a permutation of which workgroup does which row -- every row's result must be bit-identical
    to the grid-order form, lean and materialising, with empty rows and equal lengths in the mix."""
    from min_llm_inference_amd import ops
    c, d = _prepare(oracle, dev, seed, B, S, D, conditioned=True, zero_every=9)
    res = []
    try:
        for order in (1, 0):
            assert mli.mli_tune(b"scan_row_order", order) == 0
            out_lean = torch.full((B, D), SENTINEL, device=dev)
            out_full = torch.full((B, D), SENTINEL, device=dev)
            qkt = torch.full((B, S), SENTINEL, device=dev)
            ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], None, out_lean, False, phases=7, n_sequence=S)
            ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], qkt, out_full, False, phases=3)
            res.append((host(out_lean), host(out_full), host(qkt)))
    finally:
        mli.mli_tune(b"scan_row_order", 1)
    for k, what in enumerate(("lean attention_result", "materialising attention_result", "probabilities")):
        assert_equal(res[0][k], res[1][k], what=what)
    assert_equal(res[0][0], res[0][1], what="lean vs materialising")
    assert (res[0][0] != SENTINEL).all()


# ---- a decode step replayed from a hipGraph ----------------------------------------------------------------------
@pytest.mark.parametrize("bf16", [False, True])
def test_graph_replay_equals_eager_steps(mli, dev, bf16):
    """N decode steps (lean attention + fused decoder head) launched eagerly == the same N steps replayed from ONE
    captured hipGraph: lengths advance on the device, so every replay does different work through the same nodes."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from min_llm_inference_amd import ops
    n_steps = 6

    def run(graph):
        wl = bench.Workload("c3", dev, 0xABCD, headroom=n_steps + 8, dtype="bf16" if bf16 else "f32")
        toks = []
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            step = wl.lean_step
            step()                                   # warm-up: allocates the per-stream workspaces
            toks.append(wl.decoder_result.clone())
            g = ops.StepGraph(step) if graph else None
            for _ in range(n_steps):
                g.launch() if graph else step()
                toks.append(wl.decoder_result.clone())
            side.synchronize()
        return torch.stack(toks).cpu().numpy(), wl.lengths.cpu().numpy(), wl.attention_result.cpu().numpy()

    t0, l0, a0 = run(False)
    t1, l1, a1 = run(True)
    assert_equal(t1, t0, what="tokens per step")
    assert_equal(l1, l0, what="lengths")
    assert_equal(a1, a0, what="attention_result of the last step")
    assert (t0 >= 0).all() and len(np.unique(t0)) > 4


# ---- the latency-shaped small GEMM (proj_gemm_panel.hip) ----------------------------------------------------------
@pytest.mark.parametrize("seed,B,S,D", [(171, 256, 1024, 256), (172, 37, 128, 132), (173, 64, 256, 516), (174, 300, 64, 64),
                                        (175, 5, 128, 1024)])
def test_panel_projection_bit_identical_to_the_tiled_kernel(oracle, mli, dev, seed, B, S, D):
    """get_latest (paged and contiguous) through the 32x32-tile, whole-K-panel kernel == through the 64x64 tiled one,
    bit for bit (both are k-ordered fp32 fma chains), and both match the oracle; rows the op must not touch stay
    untouched.  The engines need the identity: prefill (tiled) and decode (panel) must write the same K / V rows."""
    from min_llm_inference_amd import ops
    from helpers import naive_case
    c, _ = _prepare(oracle, dev, seed, B, S, D, conditioned=True, zero_every=5)
    n = naive_case(seed + 7, B, S, D, D, conditioned=True, zero_every=5)
    n["lengths"] = np.maximum(n["lengths"] - 1, 0).astype(np.int32)   # get_latest reads position L - 1
    results = {}
    for mode in (0, 2):
        assert mli.mli_tune(b"gemm_panel", mode) == 0
        try:
            d = to_dev(c, dev)
            ops.launch_get_latest_k_q_v_paged_attention(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"],
                                                        d["q_output"], S)
            dn = {k: _t(v, dev) for k, v in n.items() if isinstance(v, np.ndarray)}
            ops.launch_get_latest_kt_q_v(dn["inp"], dn["lengths"], dn["wk"], dn["wq"], dn["wv"], dn["kt_cache"],
                                         dn["v_cache"], dn["q_output"])
            results[mode] = [host(d["pool"]).copy(), host(d["q_output"]).copy(), host(dn["kt_cache"]).copy(),
                             host(dn["v_cache"]).copy(), host(dn["q_output"]).copy()]
        finally:
            mli.mli_tune(b"gemm_panel", 1)
    for a, b, what in zip(results[0], results[2], ["page pool", "q_output (paged)", "kt_cache", "v_cache", "q_output"]):
        assert_equal(b, a, what=f"{what}: panel == tiled")
    oracle.get_latest_kt_q_v(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    assert_close(results[2][1], c["q_output"], what="q_output vs oracle")
    oracle.get_latest_kt_q_v(n["inp"], n["lengths"], n["wk"], n["wq"], n["wv"], n["kt_cache"], n["v_cache"], n["q_output"])
    assert_close(results[2][2], n["kt_cache"], what="kt_cache vs oracle")
    assert_close(results[2][3], n["v_cache"], what="v_cache vs oracle")
    assert_close(results[2][4], n["q_output"], what="q_output (contiguous) vs oracle")


@pytest.mark.parametrize("seed,B,S,D,V", [(176, 256, 128, 256, 1024), (177, 64, 128, 132, 1500), (178, 9, 64, 512, 1025)])
def test_panel_logits_and_argmax_equal_the_tiled_kernel(oracle, mli, dev, seed, B, S, D, V):
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, att, lengths = _decoder_inputs(rng, max(B, 9), S, D, V)
    emb, att, lengths = emb, att[:max(B, 9)], lengths[:max(B, 9)]
    Bn = att.shape[0]
    inp_emb = rand_f(rng, (Bn, S, D))
    out = {}
    for mode in (0, 2):
        assert mli.mli_tune(b"gemm_panel", mode) == 0
        try:
            d_inp, d_len, d_res = _t(inp_emb, dev), _t(lengths, dev), _t(np.full((Bn,), 77, np.int32), dev)
            ops.decoder_fused(_t(att, dev), _t(emb, dev), _t(wpe, dev), d_inp, d_len, d_res)
            m_inp, m_len, m_res = _t(inp_emb, dev), _t(lengths, dev), _t(np.full((Bn,), 77, np.int32), dev)
            m_score = torch.zeros(Bn, V, device=dev)
            ops.launch_decoder(_t(att, dev), _t(emb, dev), m_score, _t(wpe, dev), m_inp, m_len, m_res)
            out[mode] = [host(d_res).copy(), host(d_len).copy(), host(d_inp).copy(), host(m_score).copy(), host(m_res).copy()]
        finally:
            mli.mli_tune(b"gemm_panel", 1)
    for a, b, what in zip(out[0], out[2], ["tokens (fused)", "lengths", "inp_embedding", "emb_score", "tokens (materialising)"]):
        assert_equal(b, a, what=f"{what}: panel == tiled")
    assert_equal(out[2][0], out[2][4], what="fused == materialising tokens")
    score = np.zeros((Bn, V), np.float32)
    res = np.full((Bn,), 77, np.int32)
    L = lengths.copy()
    oracle.decoder_host(att, emb, score, wpe, inp_emb.copy(), L, res)
    assert_equal(out[2][0], res, what="tokens vs oracle")
    assert_close(out[2][3], score, what="emb_score vs oracle")


# ---- prefill: the encoder as the fill GEMM's prologue -------------------------------------------
def _prefill_inputs(rng, B, S, D, V):
    emb, wpe = rand_f(rng, (V, D)), rand_f(rng, (S, D))
    inp = rand_i(rng, (B, S), V - 1)
    lengths = rand_i(rng, (B,), S - 1)
    lengths[:4] = [0, 1, S - 1, 17]
    n_new = int(rng.integers(1, B + 1))
    new_idx = np.zeros((B,), np.int32)
    new_idx[:n_new] = rng.permutation(B)[:n_new]
    w = [((rng.random((D, D), dtype=np.float32) * 2 - 1) / np.sqrt(D)).astype(np.float32) for _ in range(2)]
    return emb, wpe, inp, lengths, n_new, new_idx, w


@pytest.mark.parametrize("compact", [1, 0])
@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("seed,B,S,D,V", [(181, 40, 256, 136, 1500), (182, 9, 1024, 512, 1024), (183, 64, 128, 2048, 1024)])
def test_paged_prefill_equals_encoder_then_fill(mli, dev, seed, B, S, D, V, bf16, compact):
    """mli_paged_prefill == mli_paged_attention_encoder + mli_fill_new_k_v_cache_paged on the whole page pool, bit for
    bit (segment 0 written once, K / V from the same fp32 resp. bf16-rounded rows), rows that are not new untouched."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, inp, lengths, n_new, new_idx, w = _prefill_inputs(rng, B, S, D, V)
    pool, table = build_page_pool(rng, lengths, S, D)
    esz = 2 if bf16 else 4

    def state():
        p = _t(bf16_bits(pool).view(np.int16), dev).view(torch.bfloat16) if bf16 else _t(pool, dev)
        return p, _t(np.where(table >= 0, p.data_ptr() + esz * table, 0).astype(np.int64), dev)

    ws = [(_t(bf16_bits(x).view(np.int16), dev).view(torch.bfloat16) if bf16 else _t(x, dev)) for x in w]
    args = (_t(emb, dev), _t(wpe, dev), _t(inp, dev))
    L, idx = _t(lengths, dev), _t(new_idx, dev)
    assert mli.mli_tune(b"fill_compact", compact) == 0
    assert mli.mli_tune(b"prefill_fused", 2) == 0   # the prologue form whatever the width (the default takes two launches beyond 512)
    try:
        p1, t1 = state()
        ops.paged_prefill(*args, t1, L, idx, ws[0], ws[1], n_new)
        mli.mli_tune(b"prefill_fused", 1)
        p3, t3 = state()
        ops.paged_prefill(*args, t3, L, idx, ws[0], ws[1], n_new)   # the form the entry point picks for this width
        p2, t2 = state()
        if bf16:
            ops.launch_paged_attention_encoder_kernel_bf16(*args, t2, L, idx, n_new)
            ops.launch_fill_new_k_v_cache_paged_attention_bf16(t2, idx, L, ws[0], ws[1], n_new, S)
        else:
            ops.launch_paged_attention_encoder_kernel(*args, t2, L, idx, n_new)
            ops.launch_fill_new_k_v_cache_paged_attention(t2, idx, L, ws[0], ws[1], n_new, S)
        torch.cuda.synchronize()
    finally:
        mli.mli_tune(b"fill_compact", 1)
        mli.mli_tune(b"prefill_fused", 1)
    view = torch.int16 if bf16 else torch.int32
    assert torch.equal(p1.view(view), p2.view(view)), "page pool: one launch == encoder + fill"
    assert torch.equal(p3.view(view), p2.view(view)), "page pool: the form picked by shape == encoder + fill"
    changed = (p2.view(view) != (_t(bf16_bits(pool).view(np.int16), dev) if bf16 else _t(pool, dev).view(view))).sum().item()
    assert changed > 0


@pytest.mark.parametrize("seed,B,S,D,V", [(184, 37, 260, 132, 1500), (185, 12, 512, 256, 1024)])
def test_prefill_contiguous_equals_encoder_then_fill(oracle, mli, dev, seed, B, S, D, V):
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, inp, lengths, n_new, new_idx, w = _prefill_inputs(rng, B, S, D, V)
    x0, kt0, v0 = rand_f(rng, (B, S, D)), rand_f(rng, (B, D, S)), rand_f(rng, (B, S, D))
    args = (_t(emb, dev), _t(wpe, dev), _t(inp, dev))
    L, idx, wk, wv = _t(lengths, dev), _t(new_idx, dev), _t(w[0], dev), _t(w[1], dev)
    x1, kt1, v1 = _t(x0, dev), _t(kt0, dev), _t(v0, dev)
    assert mli.mli_tune(b"prefill_fused", 2) == 0
    try:
        ops.prefill(*args, x1, L, idx, wk, wv, kt1, v1, n_new)
    finally:
        mli.mli_tune(b"prefill_fused", 1)
    x2, kt2, v2 = _t(x0, dev), _t(kt0, dev), _t(v0, dev)
    ops.launch_inference_optimized_encoder_kernel(*args, x2, L, idx, n_new)
    ops.launch_fill_new_kt_v_cache(x2, idx, L, wk, wv, kt2, v2, n_new)
    assert_equal(host(x1), host(x2), what="inp_embedding")
    assert_equal(host(kt1), host(kt2), what="kt_cache")
    assert_equal(host(v1), host(v2), what="v_cache")
    # and the oracle's two host functions
    oracle.inference_optimized_encoder_host(emb, wpe, inp, x0, lengths, new_idx, n_new)
    oracle.fill_new_kt_v_cache(x0, new_idx, lengths, w[0], w[1], kt0, v0, n_new)
    assert_close(host(x1), x0, thr=0, what="inp_embedding vs oracle")
    assert_close(host(kt1), kt0, what="kt_cache vs oracle")
    assert_close(host(v1), v0, what="v_cache vs oracle")


# ---- the fp32 GEMM as loader waves + MFMA waves ---------------------------------------------------------------------
@pytest.mark.parametrize("seed,B,S,D,V", [(181, 70, 64, 256, 300), (182, 33, 48, 516, 1024), (183, 130, 32, 1024, 96)])
def test_fp32_gemm_wave_split_is_bit_identical(oracle, mli, dev, seed, B, S, D, V):
    """mli_tune("gemm_split"): the 64-row-tile fp32 GEMM as 512-thread workgroups (four waves load, four multiply) against
    one wave doing both -- prefill (with and without the encoder prologue), decode projection, logits with the argmax
    epilogue and materialised logits: every output bit for bit (same MFMA chain per element)."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    res = []
    try:
        assert mli.mli_tune(b"gemm_panel", 0) == 0   # keep the small shapes on the tiled kernel
        for split in (1, 0):
            assert mli.mli_tune(b"gemm_split", split) == 0
            c, d = _prepare(oracle, dev, seed, B, S, D, conditioned=True, zero_every=6)
            ops.launch_fill_new_k_v_cache_paged_attention(d["page_table"], d["new_batch_idx"], d["lengths"], d["wk"], d["wv"],
                                                          c["n_new"], S)
            ops.launch_get_latest_k_q_v_paged_attention(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"],
                                                        d["q_output"], S)
            g = torch.Generator(device=dev); g.manual_seed(seed)
            emb = torch.rand(V, D, device=dev, generator=g) * 2 - 1
            wpe = torch.rand(S, D, device=dev, generator=g) * 2 - 1
            attn = torch.rand(B, D, device=dev, generator=g) * 2 - 1
            score = torch.zeros(B, V, device=dev)
            lengths = d["lengths"].clone()
            toks = torch.full((B, 1), -5, dtype=torch.int32, device=dev)
            ops.launch_paged_attention_decoder_multi_rounds(attn, emb, score, wpe, d["page_table"], lengths, toks, 0)
            lengths2 = d["lengths"].clone()
            toks2 = torch.full((B, 1), -5, dtype=torch.int32, device=dev)
            ops.paged_decoder_fused(attn, emb, wpe, d["page_table"], lengths2, toks2, 0, False)
            inp = torch.randint(0, V, (B, S), dtype=torch.int32, device=dev, generator=g)
            ops.paged_prefill(emb, wpe, inp, d["page_table"], d["lengths"], d["new_batch_idx"], d["wk"], d["wv"], c["n_new"])
            res.append({k: host(v) for k, v in (("pool", d["pool"]), ("q", d["q_output"]), ("score", score), ("toks", toks),
                                                ("toks_fused", toks2), ("lengths", lengths))})
    finally:
        mli.mli_tune(b"gemm_split", 1)
        mli.mli_tune(b"gemm_panel", 1)
    for k in res[0]:
        assert_equal(res[0][k], res[1][k], what=k)
    assert_equal(res[0]["toks"], res[0]["toks_fused"], what="fused head vs materialising head")


@pytest.mark.parametrize("B,V,D", [(1024, 1024, 512), (1000, 1010, 576), (960, 960, 1024), (1024, 2048, 512)])
def test_full_batch_logits_wave_split_is_bit_identical(mli, dev, B, V, D):
    """The logits GEMM of a full batch (1024 rows x 1024 vocabulary entries and neighbours: the shapes of configs 4 / 5 and of
    the reference's profiling workload) as loader waves + MFMA waves against one wave doing both (mli_tune "gemm_split" = 0):
    materialised logits, the (max, lowest index) pairs of the argmax epilogue and the tokens, bit for bit -- ragged last row /
    column tiles (rows and vocabulary entries beyond the matrix never stored, never winning), ties."""
    from min_llm_inference_amd import ops
    S = 32
    res = []
    for split in (1, 0):
        try:
            assert mli.mli_tune(b"gemm_split", split) == 0
            g = torch.Generator(device=dev); g.manual_seed(B + V + D)
            emb = torch.rand(V, D, device=dev, generator=g) * 2 - 1
            emb[V // 3] = emb[V // 7]            # two vocabulary entries with identical logits: the lower index must win
            emb[V - 1] = emb[V // 7]
            wpe = torch.rand(S, D, device=dev, generator=g) * 2 - 1
            attn = torch.rand(B, D, device=dev, generator=g) * 2 - 1
            attn[5] = emb[V // 7] * 4.0          # a row whose maximum is that tied pair
            x = torch.zeros(B, S, D, device=dev)
            score = torch.full((B, V), 9.0, device=dev)
            lengths = torch.randint(1, S - 2, (B,), dtype=torch.int32, device=dev, generator=g)
            toks = torch.full((B,), -5, dtype=torch.int32, device=dev)
            ops.launch_decoder(attn, emb, score, wpe, x, lengths.clone(), toks)
            toks2 = torch.full((B,), -5, dtype=torch.int32, device=dev)
            ops.decoder_fused(attn, emb, wpe, x.clone(), lengths.clone(), toks2)
            res.append({"score": host(score), "toks": host(toks), "toks_fused": host(toks2)})
        finally:
            mli.mli_tune(b"gemm_split", 1)
    for k in res[0]:
        assert_equal(res[0][k], res[1][k], what=f"{k}: wave split vs one wave doing both")
    assert_equal(res[0]["toks"], res[0]["toks_fused"], what="argmax epilogue vs materialised logits")
    assert int(res[0]["toks"][5]) == V // 7, "ties go to the lowest index"
    want = (host(attn).astype(np.float64) @ host(emb).astype(np.float64).T)
    assert_close(res[0]["score"], want.astype(np.float32), thr=1e-3, what="logits vs float64")
