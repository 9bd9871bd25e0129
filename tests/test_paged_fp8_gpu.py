"""GPU parity of the fp8 page extension (MLI_ELEM_FP8: OCP e4m3 x / K / V pages under the reference's layout rule, bf16
weights, fp32 q / scores / softmax / accumulation) -- SURVEY 8(f) row 4, opt-in.  PARITY UNPINNED BY THE REFERENCE: it is
fp32 only.  Expectation = the fp32 CPU oracle evaluated on what the pages hold (fp8-rounded x, K, V) and on bf16-rounded
weights, with K / V rounded to fp8 (nearest even, saturating at 448) where the kernels store them.

Stated tolerances: attention_result and q_output 1e-3 absolute against that oracle (the reference's threshold); a K / V
element the GPU stores may sit one fp8 step from the oracle's (the MFMA sums in another order, and a sum that lands on a
rounding boundary can go either way): at most 0.1 % of the elements, each by one step."""
import numpy as np
import pytest
import torch

from gpu_util import host
from helpers import (PAGE, assert_close, assert_equal, bf16_bits, bf16_round, fp8_bits, fp8_decode, fp8_round, paged_case)

pytestmark = pytest.mark.gpu
FP8 = 2   # ops.ELEM_FP8
SENTINEL = 12345.0


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _fp8_case(oracle, dev, seed, B, S, D, lengths=None, zero_every=None, poison=True):
    """A paged case whose pages hold fp8 bytes: the contiguous tensors are rounded to fp8, cloned into the pool by the
    oracle's layout rule, and the pool is uploaded as byte codes; K / V slots at or beyond a row's length are NaN (0x7f)."""
    c = paged_case(seed, B, S, D, conditioned=True, lengths=lengths, zero_every=zero_every)
    for k in ("inp_embedding", "kt_cache", "v_cache"):
        c[k] = fp8_round(c[k])
    for k in ("wk", "wq", "wv"):
        c[k] = bf16_round(c[k])
    c["pool"] = fp8_round((c["pool"] * 2 - 1).astype(np.float32))
    oracle.clone_to_pages(c["pool"], c["table"], c["inp_embedding"], c["kt_cache"], c["v_cache"], c["lengths"])
    bits = fp8_bits(c["pool"])
    if poison:   # every (row, slot >= L) of a page the row owns: K and V segments
        b_idx, s_idx = np.nonzero(np.arange(S)[None, :] >= c["lengths"][:, None])
        page = c["table"][b_idx, s_idx // PAGE]
        owned = page >= 0
        off = page[owned].astype(np.int64) + (s_idx[owned] % PAGE) * 3 * D + D
        if len(off):
            bits.reshape(-1)[(off[:, None] + np.arange(2 * D)[None, :]).reshape(-1)] = 0x7f
    d = {k: _t(v, dev) for k, v in c.items() if isinstance(v, np.ndarray) and k not in ("table", "pool", "wk", "wq", "wv")}
    d["pool"] = _t(bits, dev)
    for w in ("wk", "wq", "wv"):
        d[w] = _t(bf16_bits(c[w]).view(np.int16), dev).view(torch.bfloat16)
    d["page_table"] = _t(np.where(c["table"] >= 0, d["pool"].data_ptr() + c["table"], 0).astype(np.int64), dev)
    return c, d


def _oracle_scan(oracle, c):
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])
    return c["attention_result"]


def test_device_conversion_matches_the_numpy_model(mli, dev):
    """mli_f32_to_fp8 (clamp + v_cvt_pk_fp8_f32) against tests/helpers.fp8_bits on every rounding boundary of the
    format, values beyond +-448 (saturate; the bare instruction would give NaN from 465 on), subnormals, zeros, NaN."""
    from min_llm_inference_amd import ops
    t = fp8_decode(np.arange(127, dtype=np.uint8)).astype(np.float64)
    mids = (t[1:] + t[:-1]) / 2
    vals = np.concatenate([t, mids, np.nextafter(mids.astype(np.float32), np.float32(0)), np.nextafter(mids.astype(np.float32), np.float32(1e9)),
                           [449, 463.9, 464, 465, 480, 1e9, 3e38, 1e-9, 2.0 ** -10, 2.0 ** -11]]).astype(np.float32)
    rng = np.random.default_rng(5)
    vals = np.concatenate([vals, -vals, (rng.standard_normal(1 << 16) * 10).astype(np.float32), [np.nan, -0.0, 0.0, np.inf]]).astype(np.float32)
    vals = np.concatenate([vals, np.zeros((-len(vals)) % 4, np.float32)])
    got = host(ops.f32_to_fp8(_t(vals, dev)))
    want = fp8_bits(vals)
    nan = np.isnan(vals)
    assert ((got[nan] & 0x7f) == 0x7f).all()
    assert_equal(got[~nan], want[~nan], what="fp8 codes")


# (seed, B, S, D): rows of 4 / 2 / 1 token slots per load instruction, two lane loads per row, a width that is no power of two
SCAN_SHAPES = [(301, 24, 256, 64), (302, 9, 1024, 256), (303, 6, 4096, 512), (304, 40, 64, 512), (305, 12, 128, 1024),
               (306, 7, 2048, 2048), (307, 5, 1024, 528), (308, 300, 1024, 128), (309, 700, 128, 512), (310, 1100, 32, 512)]


@pytest.mark.parametrize("merge", [1, 0])
@pytest.mark.parametrize("seed,B,S,D", SCAN_SHAPES)
def test_fp8_lean_scan_matches_the_oracle(oracle, mli, dev, seed, B, S, D, merge):
    """The chunked single-pass scan over fp8 pages (lean form: in-kernel merge, or the separate combine launch)."""
    from min_llm_inference_amd import ops
    c, d = _fp8_case(oracle, dev, seed, B, S, D, zero_every=4)
    assert mli.mli_tune(b"scan_merge", merge) == 0
    try:
        outs = []
        for _ in range(3):   # the arrival counters must be back at zero after every launch
            d["attention_result"].fill_(SENTINEL)
            ops.decode_scan_paged(d["q_output"], d["page_table"], d["lengths"], None, d["attention_result"], FP8, phases=7,
                                  n_sequence=S)
            outs.append(host(d["attention_result"]).copy())
    finally:
        mli.mli_tune(b"scan_merge", 1)
    assert_equal(outs[1], outs[0], what="second launch")
    assert_equal(outs[2], outs[0], what="third launch")
    assert_close(outs[0], _oracle_scan(oracle, c), what="attention_result vs oracle")
    assert (outs[0][c["lengths"] == 0] == 0).all()


STREAM_CASES = [
    (321, 48, 4096, 512, None),
    (322, 300, 1024, 128, None),
    (323, 2048, 1024, 64, "short"),
    (324, 1, 4096, 256, [4095]),
    (325, 9, 1024, 1024, [0, 1023, 0, 0, 16, 17, 512, 1, 1008]),
    (326, 24, 2048, 2048, "short"),
    (327, 33, 1024, 528, None),
]


@pytest.mark.parametrize("seed,B,S,D,lengths", STREAM_CASES)
@pytest.mark.parametrize("dyn", [12, 0])
def test_fp8_lean_scan_equal_page_shares(oracle, mli, dev, seed, B, S, D, lengths, dyn):
    """attention_stream.hip over fp8 pages, forced onto small shapes: within 1e-5 of the chunked form, within 1e-3 of the
    oracle, empty rows zero, bit-identical from launch to launch."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    if isinstance(lengths, str):
        lengths = rng.integers(0, 70, size=B).astype(np.int32)
    c, d = _fp8_case(oracle, dev, seed, B, S, D, lengths=lengths, zero_every=None if lengths is not None else 7)
    q = d["q_output"]
    mli.mli_tune(b"scan_stream", 0)
    try:
        ops.decode_scan_paged(q, d["page_table"], d["lengths"], None, d["attention_result"], FP8, phases=7, n_sequence=S)
        chunked = host(d["attention_result"]).copy()
        mli.mli_tune(b"scan_stream", 1)
        assert mli.mli_tune(b"scan_stream_min_tokens", 0) == 0
        assert mli.mli_tune(b"scan_stream_dynamic_pct", dyn) == 0
        assert mli.mli_tune(b"scan_stream_granule", 16) == 0
        outs = []
        for _ in range(3):
            d["attention_result"].fill_(SENTINEL)
            ops.decode_scan_paged(q, d["page_table"], d["lengths"], None, d["attention_result"], FP8, phases=7, n_sequence=S)
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
    assert_close(outs[0], _oracle_scan(oracle, c), what="attention_result vs oracle")


def test_fp8_scan_has_no_materialising_form(mli, dev):
    from min_llm_inference_amd import MliError, ops
    B, S, D = 4, 64, 64
    z = torch.zeros(B, D, device=dev)
    with pytest.raises(MliError):
        ops.decode_scan_paged(z, torch.zeros(B, S // PAGE, dtype=torch.int64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                              torch.zeros(B, S, device=dev), z.clone(), FP8, phases=3)


def _page_rows(pool_bits, table, b, s, D):
    off = table[b, s // PAGE] + (s % PAGE) * 3 * D
    return pool_bits[off:off + D], pool_bits[off + D:off + 2 * D], pool_bits[off + 2 * D:off + 3 * D]


def _assert_fp8_rows_close(got_bits, want_vals, what, stats):
    """Stored fp8 codes against the oracle's fp32 values rounded to fp8: equal, or one fp8 step apart (counted)."""
    want_bits = fp8_bits(want_vals)
    bad = got_bits != want_bits
    stats[0] += int(bad.sum())
    stats[1] += bad.size
    if bad.any():
        g, w = fp8_decode(got_bits[bad]).astype(np.float64), fp8_decode(want_bits[bad]).astype(np.float64)
        step = np.maximum(np.abs(w), 2.0 ** -6) * 2.0 ** -3          # spacing of the format at |w| (subnormals: 2^-9)
        assert (np.abs(g - w) <= step + 1e-12).all(), what
        # ... and only where the fp32 value sits at a rounding boundary: the unrounded value is (almost) as far from its own
        # rounded value as it can be -- half a step, a quarter just below a power of two
        assert (np.abs(want_vals[bad].astype(np.float64) - w) >= 0.24 * step).all(), what


@pytest.mark.parametrize("seed,B,S,D", [(331, 37, 128, 64), (332, 24, 256, 512), (333, 6, 1024, 256), (334, 3, 4096, 512),
                                         (335, 5, 64, 2048), (336, 9, 96, 528)])
@pytest.mark.parametrize("zero_every", [None, 4])
def test_fp8_paged_attention_lean(oracle, mli, dev, seed, B, S, D, zero_every):
    """The layers' composition over fp8 pages: fill (K / V of the new rows from their fp8 x rows), latest (q, k, v of every
    row's last token), lean scan."""
    from min_llm_inference_amd import ops
    c, d = _fp8_case(oracle, dev, seed, B, S, D, zero_every=zero_every, poison=False)
    before = host(d["pool"]).copy()
    ops.paged_attention_lean(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"], d["q_output"],
                             d["attention_result"], c["n_new"], S, elem=FP8)
    oracle.fill_new_kt_v_cache(c["inp_embedding"], c["new_batch_idx"], c["lengths"], c["wk"], c["wv"], c["kt_cache"],
                               c["v_cache"], c["n_new"])
    oracle.get_latest_kt_q_v(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    assert_close(host(d["q_output"]), c["q_output"], thr=1e-4, what="q_output")
    pool = host(d["pool"])
    new_rows = set(c["new_batch_idx"][:c["n_new"]].tolist())
    stats = [0, 0]
    touched = np.zeros(pool.shape, bool)
    for b in range(B):
        L = int(c["lengths"][b])
        for s in (range(L) if b in new_rows else ([L - 1] if L else [])):
            x, k, v = _page_rows(pool, c["table"], b, s, D)
            _assert_fp8_rows_close(k, c["kt_cache"][b, :, s], f"K[{b},{s}]", stats)
            _assert_fp8_rows_close(v, c["v_cache"][b, s], f"V[{b},{s}]", stats)
            off = c["table"][b, s // PAGE] + (s % PAGE) * 3 * D
            touched[off + D:off + 3 * D] = True
    assert stats[0] <= 1e-3 * max(stats[1], 1), stats
    assert_equal(pool[~touched], before[~touched], what="bytes the composition must not touch")
    # the scan reads what the GPU stored: the oracle gets the same K / V
    c["kt_cache"] = fp8_round(c["kt_cache"])
    c["v_cache"] = fp8_round(c["v_cache"])
    assert_close(host(d["attention_result"]), _oracle_scan(oracle, c), thr=2e-3, what="attention_result vs oracle")
    assert (host(d["attention_result"])[c["lengths"] == 0] == 0).all()


def test_fp8_prefill_writes_fp8_embeddings_and_decoder_appends(oracle, mli, dev):
    """mli_paged_prefill(elem = fp8): x = fp8(emb[tok] + wpe[s]) -- fp32 sum, ONE rounding, bit-exact against numpy -- in
    segment 0, K / V of those rows from the same fp8 x; mli_paged_decoder_fused(elem = fp8) appends the next x."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(341)
    B, S, D, V = 6, 64, 80, 1030
    emb = (rng.random((V, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    emb[7] *= 600.0                                        # saturates: |x| > 448 must store +-448, not NaN
    wpe = (rng.random((S, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    wk, wv = (bf16_round(((rng.random((D, D), dtype=np.float32) * 2 - 1) / np.sqrt(D)).astype(np.float32)) for _ in range(2))
    lengths = np.array([0, 1, 16, 33, 62, 5], np.int32)
    inp = rng.integers(0, 1023, size=(B, S)).astype(np.int32)
    inp[5, :5] = 7
    npages = S // PAGE
    pool = torch.zeros(B * npages * PAGE * 3 * D, dtype=torch.uint8, device=dev)
    table = np.arange(B * npages, dtype=np.int64).reshape(B, npages) * (PAGE * 3 * D)
    ptrs = _t(pool.data_ptr() + table, dev)
    w16 = lambda a: _t(bf16_bits(a).view(np.int16), dev).view(torch.bfloat16)
    new_idx = np.array([1, 2, 3, 4, 5, 0], np.int32)
    ops.paged_prefill(_t(emb, dev), _t(wpe, dev), _t(inp, dev), ptrs, _t(lengths, dev), _t(new_idx, dev), w16(wk), w16(wv), 5,
                      elem=FP8)
    p = host(pool)
    stats = [0, 0]
    for b in range(B):
        for s in range(int(lengths[b])):
            x, k, v = _page_rows(p, table, b, s, D)
            want_x = fp8_bits(emb[inp[b, s]] + wpe[s])
            assert_equal(x, want_x, what=f"x[{b},{s}]")
            xf = fp8_decode(want_x)
            _assert_fp8_rows_close(k, (xf.astype(np.float32) @ wk).astype(np.float32), f"K[{b},{s}]", stats)
            _assert_fp8_rows_close(v, (xf.astype(np.float32) @ wv).astype(np.float32), f"V[{b},{s}]", stats)
    assert (p.reshape(B, -1)[0] == 0).all()                # the empty row's pages are untouched
    assert np.isfinite(fp8_decode(p)).all()
    assert (np.abs(fp8_decode(_page_rows(p, table, 5, 0, D)[0])) == 448).sum() >= D // 8   # (|x| > 448 saturates)
    # decoder: next embedding of each live row at position lengths[b]
    att = (rng.random((B, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    d_len = _t(lengths.copy(), dev)
    res = torch.full((B, 1), 7, dtype=torch.int32, device=dev)
    ops.paged_decoder_fused(_t(att, dev), _t(emb, dev), _t(wpe, dev), ptrs, d_len, res, 0, FP8)
    toks = host(res).ravel()
    exp = (att.astype(np.float64) @ emb.astype(np.float64).T).argmax(1)
    p = host(pool)
    assert toks[0] == -1
    for b in range(1, B):
        assert toks[b] == exp[b]
        L = int(lengths[b])
        x, _, _ = _page_rows(p, table, b, L, D)
        assert_equal(x, fp8_bits(emb[toks[b]] + wpe[L]), what=f"next x[{b}]")
    assert_equal(host(d_len), np.array([0, 2, 17, 34, 63, 6], np.int32))


def test_fp8_engine_matches_cpu_engine_on_fp8_rounded_state(oracle, mli, dev):
    """MLI_ENGINE_PAGED_FP8 end to end (pipelined loop, tight pool: growth and preemption) against the CPU engine with
    bf16-rounded weights and fp8-rounded page contents.  A stored K / V element can land on the other side of a rounding
    boundary and a near-tie argmax can then flip, so the bar is the bf16 engine's: every item finishes with its prompt
    intact and at least 85 % of the items are token-identical."""
    from engine_sim import make_items, make_model, run_cpu_engine
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 128, 64, 1024
    model = make_model(349, V, S, D)
    items = make_items(350, 40, 1, 60)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S, bf16="fp8")
    outs = []
    for rounds, n_blocks, pipelined in ((1, 4 * B, True), (2, 8 * B, False)):
        e = eng.Engine(eng.PAGED_FP8, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                       n_blocks=n_blocks, n_forward_rounds=rounds)
        e.set_pipelined(pipelined)
        for item_id, toks in items:
            e.add_item(item_id, toks)
        st = e.run()
        got = {i: t for i, t in e.finished()}
        e.close()
        assert st.finished == len(items)
        same = 0
        for item_id, toks in items:
            assert (got[item_id][:len(toks)] == toks).all()
            assert len(got[item_id]) == S or got[item_id][-1] == 1023
            same += len(got[item_id]) == len(cpu[item_id]) and bool((got[item_id] == cpu[item_id]).all())
        assert same >= 0.85 * len(items), same
        outs.append(got)
    # scheduling (rounds, pool size, loop order, preemption) never changes an item's tokens
    for item_id, _ in items:
        assert len(outs[0][item_id]) == len(outs[1][item_id]) and (outs[0][item_id] == outs[1][item_id]).all(), item_id
