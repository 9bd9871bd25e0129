"""The row-sharded engine group below Python (include/mli_shard.h): one engine per GPU inside one process, one RCCL
all-gather of the generated token ids per rank and iteration.  A GPU test box has ONE GPU, so what runs here is the group of
one rank -- the same code path (ncclCommInitAll, the all-reduce of ones, ncclAllGather on the engine's stream) over a 1-rank
communicator; N > 1 is unmeasured.  The reference has no multi-GPU code to compare with (its README lists it as a plan):
the bar is that sharding changes nothing -- every item's tokens equal the single engine's."""
import ctypes

import numpy as np
import pytest

from engine_sim import make_items, make_model

pytestmark = pytest.mark.gpu


def _single(kind, model, items, B, S, n_blocks):
    from min_llm_inference_amd import engine as eng
    D, V = model["wk"].shape[0], model["emb_table"].shape[0]
    e = eng.Engine(kind, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                   n_blocks=n_blocks)
    e.set_pipelined(False)
    for item_id, toks in items:
        e.add_item(item_id, toks)
    e.run()
    out = {i: t for i, t in e.finished()}
    e.close()
    return out


@pytest.mark.parametrize("kind_name", ["PAGED_GEMM", "PAGED_BF16"])
def test_group_of_one_rank_generates_the_single_engines_tokens(mli, dev, kind_name):
    import torch
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 32, 128, 128, 1024
    kind = getattr(eng, kind_name)
    model = make_model(91, V, S, D)
    items = make_items(92, 3 * B, 1, 50)
    want = _single(kind, model, items, B, S, 4 * B)

    g = eng.ShardGroup(kind, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                       devices=[0], n_blocks=4 * B)
    for item_id, toks in items:
        g.add_item(item_id, toks)
    st = g.run()
    assert st.ranks_seen == 1 and st.finished == len(items) and st.iterations > 0
    e0 = g.engine(0)
    got = {i: t for i, t in e0.finished()}
    assert st.total_tokens == sum(len(got[i]) - len(t) for i, t in items)
    for item_id, _ in items:
        assert len(got[item_id]) == len(want[item_id]) and (got[item_id] == want[item_id]).all(), item_id
    # the gathered buffer of the last iteration holds what the rank's decoder wrote
    gp, gn = g.gathered_ptr(0)
    rp, rn = e0.decoder_result_ptr()
    assert gn == rn == B
    torch.cuda.synchronize()
    host = np.empty((2, B), np.int32)
    hip = ctypes.CDLL("libamdhip64.so")
    for row, ptr in enumerate((gp, rp)):
        assert hip.hipMemcpy(host[row].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), B * 4, 2) == 0
    assert (host[0] == host[1]).all()
    g.close()


def test_duplicate_devices_and_bad_arguments_are_refused(mli, dev):
    from min_llm_inference_amd import MliError, engine as eng
    B, S, D, V = 8, 64, 64, 1024
    model = make_model(93, V, S, D)
    with pytest.raises(MliError, match="duplicate device"):
        eng.ShardGroup(eng.PAGED, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                       devices=[0, 0], n_blocks=4 * B)
    with pytest.raises(MliError):
        eng.ShardGroup(eng.PAGED, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                       devices=[], n_blocks=4 * B)


@pytest.mark.parametrize("n_ranks,n_items", [(2, 90), (3, 40), (4, 2)])
def test_loopback_group_runs_ranks_in_lock_step(mli, dev, n_ranks, n_items):
    """N ranks on the one GPU of the test box (mli_shard_group_create_loopback: the exchange is device-to-device copies, everything
    else -- item dealing, one host thread per rank, the lock-step barrier, ranks that run out of work early or never had any
    (4 ranks, 2 items) -- is the code of the RCCL group).  Every item's tokens equal the single engine's; the gathered buffer of
    every rank holds every rank's last decoder output."""
    import torch
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 128, 128, 1024
    model = make_model(94, V, S, D)
    items = make_items(95, n_items, 1, 50)
    want = _single(eng.PAGED_GEMM, model, items, B * n_ranks, S, 4 * B * n_ranks)
    g = eng.ShardGroup(eng.PAGED_GEMM, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                       devices=[0], n_blocks=4 * B, loopback_ranks=n_ranks)
    for item_id, toks in items:
        g.add_item(item_id, toks)
    st = g.run()
    assert st.ranks_seen == n_ranks and st.finished == len(items)
    got = {}
    for r in range(n_ranks):
        mine = {i: t for i, t in g.engine(r).finished()}
        assert all(i % n_ranks == r for i in mine), "items are dealt to rank id % n_ranks"
        got.update(mine)
    assert st.total_tokens == sum(len(got[i]) - len(t) for i, t in items)
    for item_id, _ in items:
        assert len(got[item_id]) == len(want[item_id]) and (got[item_id] == want[item_id]).all(), item_id
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    results = np.empty((n_ranks, B), np.int32)
    for r in range(n_ranks):
        rp, rn = g.engine(r).decoder_result_ptr()
        assert rn == B and hip.hipMemcpy(results[r].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(rp), B * 4, 2) == 0
    for r in range(n_ranks):
        gp, gn = g.gathered_ptr(r)
        assert gn == n_ranks * B
        buf = np.empty((n_ranks, B), np.int32)
        assert hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(gp), n_ranks * B * 4, 2) == 0
        # a rank that stopped stepping keeps its last output; the copies of the last common iteration carry it everywhere
        assert (buf == results).all(), f"gathered buffer of rank {r}"
    g.close()
