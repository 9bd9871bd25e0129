"""GPU parity of the encoder (new rows only) and the greedy decoder head, contiguous and paged, against the oracle.
Mirrors the reference's tests/encoder_test.cpp:35-109 and tests/decoder_test.cpp:8-341 (incl. the L = S-1 and L = 0
edge rows of MaxLengthTest) on seeded inputs; whole tensors / the whole page pool are compared, so untouched
bytes must stay untouched."""
import numpy as np
import pytest
import torch

from gpu_util import host
from helpers import assert_close, assert_equal, build_page_pool, rand_f, rand_i, scatter_rows_to_pool

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("seed,B,S,D,V", [(51, 37, 260, 132, 1500), (52, 128, 64, 512, 1024), (53, 5, 1024, 64, 3000)])
def test_inference_optimized_encoder(oracle, mli, dev, seed, B, S, D, V):
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe = rand_f(rng, (V, D)), rand_f(rng, (S, D))
    inp = rand_i(rng, (B, S), V - 1)
    out = rand_f(rng, (B, S, D))
    lengths = rand_i(rng, (B,), S)
    lengths[:2] = [0, S]
    n_new = int(rng.integers(1, B + 1))
    new_idx = np.zeros((B,), np.int32)
    new_idx[:n_new] = rng.permutation(B)[:n_new]
    d_out = _t(out, dev)
    ops.launch_inference_optimized_encoder_kernel(_t(emb, dev), _t(wpe, dev), _t(inp, dev), d_out, _t(lengths, dev),
                                                  _t(new_idx, dev), n_new)
    oracle.inference_optimized_encoder_host(emb, wpe, inp, out, lengths, new_idx, n_new)
    assert_close(host(d_out), out, thr=0, what="inp_embedding (exact: one fp32 add per element)")


@pytest.mark.parametrize("seed,B,S,D,V", [(54, 40, 256, 132, 1500), (55, 9, 1024, 512, 1024)])
def test_paged_attention_encoder(oracle, mli, dev, seed, B, S, D, V):
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe = rand_f(rng, (V, D)), rand_f(rng, (S, D))
    inp = rand_i(rng, (B, S), V - 1)
    lengths = rand_i(rng, (B,), S - 1)
    lengths[:3] = [0, 1, S - 1]
    pool, table = build_page_pool(rng, lengths, S, D)
    n_new = int(rng.integers(1, B + 1))
    new_idx = np.zeros((B,), np.int32)
    new_idx[:n_new] = rng.permutation(B)[:n_new]
    d_pool = _t(pool, dev)
    ptrs = _t(np.where(table >= 0, d_pool.data_ptr() + 4 * table, 0).astype(np.int64), dev)
    ops.launch_paged_attention_encoder_kernel(_t(emb, dev), _t(wpe, dev), _t(inp, dev), ptrs, _t(lengths, dev),
                                              _t(new_idx, dev), n_new)
    rows = [(int(b), s) for b in new_idx[:n_new] for s in range(int(lengths[b]))]
    expect = pool.copy()
    if rows:
        bb = np.array([r[0] for r in rows]); ss = np.array([r[1] for r in rows])
        scatter_rows_to_pool(expect, table, rows, 0, emb[inp[bb, ss]] + wpe[ss])
    assert_equal(host(d_pool), expect, what="page pool after the paged encoder")


def _decoder_inputs(rng, B, S, D, V):
    emb = (rng.random((V, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    wpe = rand_f(rng, (S, D))
    att = (rng.random((B, D), dtype=np.float32) * 2 - 1).astype(np.float32)
    lengths = rand_i(rng, (B,), S - 2)
    lengths[:4] = [0, S - 1, S - 2, 1]            # empty slot, finishes on length, last writable position, first token
    emb[1023] = 0
    att[5] = 1.0
    emb[1023, :] = 0.5                            # row 5 is pushed onto EOF below
    emb[1023] *= 4
    return emb, wpe, att, lengths


@pytest.mark.parametrize("seed,B,S,D,V", [(56, 64, 128, 132, 1500), (57, 200, 64, 512, 1024), (58, 8, 4096, 64, 1024)])
def test_decoder_contiguous(oracle, mli, dev, seed, B, S, D, V):
    """reference DecoderKernelTest + MaxLengthTest (tests/decoder_test.cpp:8-94)."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, att, lengths = _decoder_inputs(rng, B, S, D, V)
    inp_emb = rand_f(rng, (B, S, D))
    score = np.zeros((B, V), np.float32)
    res = np.full((B,), 77, np.int32)
    d_inp, d_len, d_res, d_score = _t(inp_emb, dev), _t(lengths, dev), _t(res, dev), _t(score, dev)
    ops.launch_decoder(_t(att, dev), _t(emb, dev), d_score, _t(wpe, dev), d_inp, d_len, d_res)
    oracle.decoder_host(att, emb, score, wpe, inp_emb, lengths, res)
    assert_close(host(d_score), score, what="emb_score")
    assert_equal(host(d_res), res, what="decoder_result")
    assert_equal(host(d_len), lengths, what="lengths")
    assert_equal(host(d_inp), inp_emb, what="inp_embedding (next token rows; everything else untouched)")
    assert res[0] == -1 and lengths[0] == 0 and lengths[1] == 0 and res[5] == 1023 and lengths[5] == 0


@pytest.mark.parametrize("variant", ["plain", "cublas"])
@pytest.mark.parametrize("seed,B,S,D,V", [(59, 64, 128, 132, 1500), (60, 100, 256, 512, 1024)])
def test_decoder_paged_multi_rounds(oracle, mli, dev, seed, B, S, D, V, variant):
    """reference PagedAttention[Cublas]DecoderKernelTest + ...MaxLengthTest (tests/decoder_test.cpp:96-341); the
    decoder_result column i_decoder of [B, n_rounds] is written, other columns untouched."""
    from min_llm_inference_amd import ops
    rng = np.random.default_rng(seed)
    emb, wpe, att, lengths = _decoder_inputs(rng, B, S, D, V)
    pool, table = build_page_pool(rng, lengths, S, D)
    n_rounds, i_dec = 3, 1
    score = np.zeros((B, V), np.float32)
    res = np.full((B, n_rounds), 77, np.int32)
    d_pool = _t(pool, dev)
    ptrs = _t(np.where(table >= 0, d_pool.data_ptr() + 4 * table, 0).astype(np.int64), dev)
    d_len, d_res, d_score = _t(lengths, dev), _t(res, dev), _t(score, dev)
    fn = (ops.launch_paged_attention_decoder_multi_rounds if variant == "plain"
          else ops.launch_paged_attention_cublas_decoder_multi_rounds)
    fn(_t(att, dev), _t(emb, dev), d_score, _t(wpe, dev), ptrs, d_len, d_res, i_dec)
    oracle.paged_decoder_host(att, emb, score, wpe, pool, table, lengths, res, i_dec)
    assert_close(host(d_score), score, what="emb_score")
    assert_equal(host(d_res), res, what="decoder_result")
    assert_equal(host(d_len), lengths, what="lengths")
    assert_equal(host(d_pool), pool, what="page pool (next embeddings in segment 0; everything else untouched)")
