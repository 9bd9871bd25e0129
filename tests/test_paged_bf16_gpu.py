"""GPU parity of the bf16 paged path (BASELINE config 4 dtype).  PARITY UNPINNED BY THE REFERENCE: it is fp32
only.  Expectation = the fp32 CPU oracle evaluated on bf16-rounded inputs, with K/V rounded to bf16 where the
kernels store them.  Products of two bf16 values are exact in fp32 and the MFMA accumulates in the oracle's k
order, so q_output and the K/V written into the pages are compared BIT-EXACTLY; scores / probabilities /
attention_result, whose sums run in a different order, within 1e-3 absolute (conditioned data, |values| ~ 1)."""
import numpy as np
import pytest
import torch

from gpu_util import host
from helpers import assert_close, assert_equal, bf16_bits, bf16_round, paged_case

pytestmark = pytest.mark.gpu

SHAPES = [(61, 37, 128, 64), (62, 24, 256, 512), (63, 6, 1024, 256), (64, 3, 4096, 512), (65, 5, 64, 2048), (66, 9, 96, 520), (67, 2, 48, 2560)]


def _case(oracle, dev, seed, B, S, D, zero_every=None):
    c = paged_case(seed, B, S, D, conditioned=True, zero_every=zero_every)
    for k in ("inp_embedding", "kt_cache", "v_cache", "wk", "wq", "wv", "pool"):
        c[k] = bf16_round(c[k])
    oracle.clone_to_pages(c["pool"], c["table"], c["inp_embedding"], c["kt_cache"], c["v_cache"], c["lengths"])
    d = {k: torch.from_numpy(v.copy()).to(dev) for k, v in c.items()
         if isinstance(v, np.ndarray) and k not in ("table", "pool", "wk", "wq", "wv")}
    d["pool"] = torch.from_numpy(bf16_bits(c["pool"]).view(np.int16)).to(dev)  # raw bf16 bits
    for w in ("wk", "wq", "wv"):
        d[w] = torch.from_numpy(bf16_bits(c[w]).view(np.int16)).to(dev)
    ptrs = np.where(c["table"] >= 0, d["pool"].data_ptr() + 2 * c["table"], 0).astype(np.int64)
    d["page_table"] = torch.from_numpy(ptrs).to(dev)
    return c, d


def _bits_within_one_ulp(got, want):
    """bf16 values equal up to one bf16 ulp (2^-7 relative) plus the fp32 reassociation error of a 2048-term sum
    (matters only where cancellation leaves a result near zero)."""
    g = (got.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    w = (want.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return (np.abs(g - w) <= 2.0 ** -7 * np.abs(w) + 1e-5).all()


@pytest.mark.parametrize("native,fused", [(1, 1), (0, 0), (1, 0)])
@pytest.mark.parametrize("zero_every", [None, 4])
@pytest.mark.parametrize("seed,B,S,D", SHAPES)
def test_paged_attention_bf16(oracle, mli, dev, seed, B, S, D, zero_every, native, fused):
    """native=1: v_mfma_f32_32x32x16_bf16 (the hardware sums the 16 products of a step in its own order, so q and
    the stored K/V may differ from the sequential fp32 oracle by fp32 rounding: q within 1e-4, K/V bits within one
    bf16 ulp).  native=0: fp32-widened operands, bit-exact."""
    from min_llm_inference_amd import ops
    assert mli.mli_tune(b"bf16_native_mfma", native) == 0
    assert mli.mli_tune(b"fused_softmax", fused) == 0
    c, d = _case(oracle, dev, seed, B, S, D, zero_every)
    ops.paged_attention_bf16(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"], d["new_batch_idx"],
                             d["q_output"], d["qkt_output"], d["attention_result"], c["n_new"], S)
    # oracle: fp32 on the bf16-rounded inputs; K/V are stored in bf16
    oracle.fill_new_kt_v_cache(c["inp_embedding"], c["new_batch_idx"], c["lengths"], c["wk"], c["wv"], c["kt_cache"],
                               c["v_cache"], c["n_new"])
    oracle.get_latest_kt_q_v(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    c["kt_cache"] = bf16_round(c["kt_cache"])
    c["v_cache"] = bf16_round(c["v_cache"])
    oracle.qkt_host(c["q_output"], c["kt_cache"], c["lengths"], c["qkt_output"])
    oracle.softmax_in_place_with_lengths_host(c["qkt_output"], c["lengths"])
    oracle.softmax_v_host(c["qkt_output"], c["v_cache"], c["attention_result"], c["lengths"])

    mli.mli_tune(b"bf16_native_mfma", 1)
    mli.mli_tune(b"fused_softmax", -1)
    if native:
        assert_close(host(d["q_output"]), c["q_output"], thr=1e-4, what="q_output")
    else:
        assert_equal(host(d["q_output"]), c["q_output"], what="q_output (bit exact)")
    pool = host(d["pool"]).view(np.uint16)
    new_rows = set(c["new_batch_idx"][:c["n_new"]].tolist())
    for b in range(B):
        L = int(c["lengths"][b])
        for s in (range(L) if b in new_rows else ([L - 1] if L else [])):
            off = c["table"][b, s // 16] + (s % 16) * 3 * D
            if native:
                assert _bits_within_one_ulp(pool[off + D:off + 2 * D], bf16_bits(c["kt_cache"][b, :, s])), f"K[{b},{s}]"
                assert _bits_within_one_ulp(pool[off + 2 * D:off + 3 * D], bf16_bits(c["v_cache"][b, s])), f"V[{b},{s}]"
            else:
                assert_equal(pool[off + D:off + 2 * D], bf16_bits(c["kt_cache"][b, :, s]), what=f"K[{b},{s}] bits")
                assert_equal(pool[off + 2 * D:off + 3 * D], bf16_bits(c["v_cache"][b, s]), what=f"V[{b},{s}] bits")
    # a K/V element one bf16 ulp off moves a score by <= 2^-8 * |q_d * k_d| ~ 1e-3 at these magnitudes
    tol = 5e-3 if native else 1e-3
    assert_close(host(d["qkt_output"]), c["qkt_output"], thr=tol, what="probabilities")
    assert_close(host(d["attention_result"]), c["attention_result"], thr=tol, what="attention_result")


def test_bf16_encoder_decoder_write_bf16_embeddings(oracle, mli, dev):
    """x = emb[tok] + wpe[s] is summed in fp32 and rounded once to bf16 in the page (encoder and decoder)."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(67)
    B, S, D, V = 5, 64, 72, 1030
    emb = (rng.random((V, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    wpe = (rng.random((S, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    lengths = np.array([0, 1, 16, 33, 62], np.int32)
    inp = rng.integers(0, 1023, size=(B, S)).astype(np.int32)
    npages = S // 16
    pool = torch.zeros(B * npages * 16 * 3 * D, dtype=torch.int16, device=dev)
    table = np.arange(B * npages, dtype=np.int64).reshape(B, npages) * (16 * 3 * D)
    ptrs = torch.from_numpy(pool.data_ptr() + 2 * table).to(dev)
    t = lambda a: torch.from_numpy(a).to(dev)
    new_idx = np.array([1, 2, 3, 4, 0], np.int32)
    ops.launch_paged_attention_encoder_kernel_bf16(t(emb), t(wpe), t(inp), ptrs, t(lengths), t(new_idx), 4)
    p = host(pool).view(np.uint16)
    for b in range(B):
        for s in range(int(lengths[b])):
            off = table[b, s // 16] + (s % 16) * 3 * D
            assert_equal(p[off:off + D], bf16_bits(emb[inp[b, s]] + wpe[s]), what=f"x[{b},{s}]")
    assert (p.reshape(B, -1)[0] == 0).all()  # empty row untouched
    # decoder: next embedding of each live row at position lengths[b]
    att = (rng.random((B, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    d_len = t(lengths.copy())
    res = torch.full((B, 1), 7, dtype=torch.int32, device=dev)
    ops.launch_paged_attention_decoder_multi_rounds_bf16(t(att), t(emb), torch.zeros(B, V, device=dev), t(wpe), ptrs,
                                                         d_len, res, 0)
    toks = host(res).ravel()
    exp = (att.astype(np.float64) @ emb.astype(np.float64).T).argmax(1)
    p = host(pool).view(np.uint16)
    assert toks[0] == -1
    for b in range(1, B):
        assert toks[b] == exp[b]
        L = int(lengths[b])
        off = table[b, L // 16] + (L % 16) * 3 * D
        assert_equal(p[off:off + D], bf16_bits(emb[toks[b]] + wpe[L]), what=f"next x[{b}]")
    assert_equal(host(d_len), np.array([0, 2, 17, 34, 63], np.int32))


@pytest.mark.parametrize("seed,B,S,D", [(68, 200, 64, 256), (69, 150, 32, 136)])
def test_bf16_latest_tall_tiles_equal_square_tiles(oracle, mli, dev, seed, B, S, D):
    """The 128x64x64 tile of the native bf16 MFMA kernel (large batches) against the 64x64x32 one.  Both sum the 16
    products of an MFMA step in hardware order and the steps in k order, so the results are bit-identical."""
    from min_llm_inference_amd import ops
    got = []
    try:
        for tall in (2, 0):
            assert mli.mli_tune(b"gemm_tall_tiles", tall) == 0
            c, d = _case(oracle, dev, seed, B, S, D, zero_every=4)
            ops.launch_get_latest_k_q_v_paged_attention_bf16(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"],
                                                             d["q_output"], S)
            got.append((host(d["pool"]), host(d["q_output"])))
    finally:
        mli.mli_tune(b"gemm_tall_tiles", 1)
    assert_equal(got[0][0], got[1][0], what="bf16 page pool: 128-row vs 64-row tiles")
    assert_equal(got[0][1], got[1][1], what="q_output: 128-row vs 64-row tiles")
    # and against the oracle (fp32 math on bf16-rounded inputs)
    oracle.get_latest_kt_q_v(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    assert_close(got[0][1], c["q_output"], thr=1e-4, what="q_output vs oracle")


@pytest.mark.parametrize("seed,B,S,D", [(70, 200, 32, 1024), (71, 150, 32, 1152), (72, 300, 32, 2048), (73, 400, 16, 1088)])
def test_bf16_latest_loader_mfma_wave_split_equals_the_tiled_kernel(oracle, mli, dev, seed, B, S, D):
    """The kernel of the large bf16 decode projection (LDS-DMA loader waves + MFMA waves, 128 x 192 tiles over the [Wk | Wq | Wv]
    column sequence: tiles that straddle two weights, XCD-aware and linear tile order, three LDS stages, three fragment sets)
    against the 128 x 64 tiled kernel.  Same MFMA steps in the same k order, so pages and q_output are bit-identical -- empty
    rows, a ragged last row tile, k extents of every residue mod 3 (the loop body is three tiles) included."""
    from min_llm_inference_amd import ops
    got = []
    try:
        assert mli.mli_tune(b"gemm_tall_tiles", 2) == 0   # the large-batch kernels whatever the batch
        for split in (1, 0):
            assert mli.mli_tune(b"gemm_bf16_split", split) == 0
            c, d = _case(oracle, dev, seed, B, S, D, zero_every=4)
            ops.launch_get_latest_k_q_v_paged_attention_bf16(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"],
                                                             d["q_output"], S)
            got.append((host(d["pool"]), host(d["q_output"])))
    finally:
        mli.mli_tune(b"gemm_tall_tiles", 1)
        mli.mli_tune(b"gemm_bf16_split", 1)
    assert_equal(got[0][0], got[1][0], what="bf16 page pool: loader / MFMA wave kernel vs tiled")
    assert_equal(got[0][1], got[1][1], what="q_output: loader / MFMA wave kernel vs tiled")
    oracle.get_latest_kt_q_v(c["inp_embedding"], c["lengths"], c["wk"], c["wq"], c["wv"], c["kt_cache"], c["v_cache"],
                             c["q_output"])
    assert_close(got[0][1], c["q_output"], thr=2e-4, what="q_output vs oracle")


@pytest.mark.parametrize("B,S,D", [(1024, 16, 2048), (300, 16, 1088)])
def test_bf16_lds_dma_projection_is_stable_over_many_launches_under_load(oracle, mli, dev, B, S, D):
    """Race screen of the LDS-DMA projection kernel (counted vmcnt, raw barriers, stages re-used two tiles later): 60 launches
    while a second stream streams 512 MiB through the chip (DMA landings arrive late and out of step), every one bit-identical to
    the tiled kernel's pages and q_output.  tools/gemm_bf16_race_screen.py runs the long form (profiles/r03_gemm_bf16_race_screen.json)."""
    from min_llm_inference_amd import ops
    c, d = _case(oracle, dev, 75, B, S, D, zero_every=6)
    pool0 = d["pool"].clone()
    side = torch.cuda.Stream(device=dev)
    big, sink = torch.empty(1 << 27, device=dev).uniform_(), torch.zeros(64, device=dev)

    def run(split):
        assert mli.mli_tune(b"gemm_bf16_split", split) == 0
        d["pool"].copy_(pool0)
        d["q_output"].fill_(3.0)
        ops.launch_get_latest_k_q_v_paged_attention_bf16(d["page_table"], d["lengths"], d["wk"], d["wq"], d["wv"], d["q_output"], S)
        return d["pool"].clone(), d["q_output"].clone()

    try:
        assert mli.mli_tune(b"gemm_tall_tiles", 2) == 0
        p_ref, q_ref = run(0)
        for i in range(60):
            if i % 4 == 0:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    ops.stream_read(big, sink)
            p, q = run(1)
            assert torch.equal(p, p_ref) and torch.equal(q, q_ref), f"launch {i} differs from the tiled kernel"
    finally:
        mli.mli_tune(b"gemm_tall_tiles", 1)
        mli.mli_tune(b"gemm_bf16_split", 1)
        torch.cuda.synchronize()
