"""GPU engine tests through the engine C ABI (include/mli_engine.h).

Mirrors the reference's tests/inferencer_test.cpp:12-285 and paged_attention_vs_naive_attention_test.cpp
(run the engines to completion, finish_count == n_items) and adds what the reference's
"Compare2Inferences" set out to do but does not (it compares a list with itself): the contiguous, paged
and paged-GEMM engines -- with a pool small enough to force page growth and preemption -- must generate
the same tokens per item, and the same tokens as the CPU engine built from the oracle."""
import numpy as np
import pytest

from engine_sim import make_items, make_model, run_cpu_engine

pytestmark = pytest.mark.gpu


def _run(kind, model, items, B, S, n_blocks=0, rounds=1, quirk=False, pipelined=False):
    from min_llm_inference_amd import engine as eng
    D = model["wk"].shape[0]
    V = model["emb_table"].shape[0]
    e = eng.Engine(kind, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                   n_blocks=n_blocks, n_forward_rounds=rounds, reference_length_reset_quirk=quirk)
    if kind != eng.CONTIGUOUS:
        e.set_pipelined(bool(pipelined))   # explicit either way: the default picks the pipelined loop where it applies
    for item_id, toks in items:
        e.add_item(item_id, toks)
    st = e.run()
    out = {i: t for i, t in e.finished()}
    e.close()
    return st, out


def test_contiguous_engine_finishes(mli, dev):
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 48, 256, 260, 1500
    model = make_model(41, V, S, D)
    items = make_items(42, 2 * B + 7, 1, S // 2)
    st, out = _run(eng.CONTIGUOUS, model, items, B, S)
    assert st.finished == len(items) and st.waiting == 0 and st.in_flight == 0
    assert st.total_tokens == sum(len(out[i]) - len(t) for i, t in items)


@pytest.mark.parametrize("kind_name", ["PAGED", "PAGED_GEMM"])
def test_paged_engine_finishes_with_growth_and_preemption(mli, dev, kind_name):
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 64, 160, 256, 1024
    model = make_model(43, V, S, D)
    items = make_items(44, 2 * B, 1, 63)
    # 4 pages per slot, like the reference (tests/inferencer_test.cpp:47): rows outgrow them -> growth + preemption
    st, out = _run(getattr(eng, kind_name), model, items, B, S, n_blocks=4 * B)
    assert st.finished == len(items)
    for i, t in items:
        assert (out[i][:len(t)] == t).all()
        assert len(out[i]) == S or out[i][-1] == 1023


def test_engines_agree_token_for_token(oracle, mli, dev):
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 128, 64, 1024
    model = make_model(45, V, S, D)
    items = make_items(46, 40, 1, 60)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    _, naive = _run(eng.CONTIGUOUS, model, items, B, S)
    _, paged = _run(eng.PAGED, model, items, B, S, n_blocks=4 * B)          # tight pool: preemption happens
    _, roomy = _run(eng.PAGED_GEMM, model, items, B, S, n_blocks=8 * B)     # S/16 pages per slot: never preempts
    _, multi = _run(eng.PAGED, model, items, B, S, n_blocks=8 * B, rounds=4)
    for item_id, _ in items:
        for name, got in (("contiguous", naive), ("paged", paged), ("paged_gemm", roomy), ("paged 4 rounds", multi)):
            assert len(got[item_id]) == len(cpu[item_id]) and (got[item_id] == cpu[item_id]).all(), (name, item_id)


def test_config1_engine_matches_cpu_path(oracle, mli, dev):
    """BASELINE config 1: B=4, D=64, S=128."""
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 4, 128, 64, 1024
    model = make_model(31, V, S, D)
    items = make_items(32, 10, 1, 40)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    st, got = _run(eng.CONTIGUOUS, model, items, B, S)
    assert st.finished == len(items)
    for item_id, _ in items:
        assert (got[item_id] == cpu[item_id]).all()


def test_reference_length_reset_quirk_still_finishes(mli, dev):
    """With the reference's stale-length upload reproduced, items still finish (host-side counting) but rows
    attend over their prompt only; kept measurable, off by default."""
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 32, 128, 64, 1024
    model = make_model(47, V, S, D)
    items = make_items(48, 2 * B, 1, 40)
    st, _ = _run(eng.PAGED, model, items, B, S, n_blocks=4 * B, quirk=True)
    assert st.finished == len(items)


def test_bf16_engine_matches_cpu_engine_on_bf16_rounded_state(oracle, mli, dev):
    """EXTENSION, parity unpinned by the reference (fp32 only).  Expectation = the CPU engine with Wk/Wq/Wv, input
    embeddings, K and V rounded to bfloat16 where the GPU stores them.  With the operands widened to the exact fp32
    MFMA (bf16_native_mfma = 0) K/V bits equal the CPU's, so tokens must agree item for item; the native bf16 MFMA
    accumulates in a different order, a K/V element can round the other way and a near-tie argmax can flip, so there
    the bar is: every item finishes with its prompt intact and at least 90 % of the items are token-identical."""
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 128, 64, 1024
    model = make_model(49, V, S, D)
    items = make_items(50, 40, 1, 60)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S, bf16=True)
    try:
        assert mli.mli_tune(b"bf16_native_mfma", 0) == 0
        _, exact = _run(eng.PAGED_BF16, model, items, B, S, n_blocks=4 * B)   # tight pool: preemption happens
    finally:
        mli.mli_tune(b"bf16_native_mfma", 1)
    for item_id, _ in items:
        assert len(exact[item_id]) == len(cpu[item_id]) and (exact[item_id] == cpu[item_id]).all(), item_id
    st, native = _run(eng.PAGED_BF16, model, items, B, S, n_blocks=8 * B, rounds=2)
    assert st.finished == len(items)
    same = 0
    for item_id, toks in items:
        assert (native[item_id][:len(toks)] == toks).all()
        same += len(native[item_id]) == len(cpu[item_id]) and bool((native[item_id] == cpu[item_id]).all())
    assert same >= 0.9 * len(items), same


def test_two_engines_on_private_streams_overlap_safely(oracle, mli, dev):
    """Two engines in one process, one host thread each, each on its own non-blocking stream (so their kernels and
    copies interleave on the GPU): per-engine scratch, counters and stream-ordered copies must keep every item's
    token stream equal to the single-engine / CPU result."""
    import threading
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 4096 // 16, 512, 1024      # S = 256, D = 512: split-sequence scratch in use
    model = make_model(51, V, S, D)
    items = make_items(52, 48, 1, 100)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    engines, outs, errs = [], [None, None], []
    for r in range(2):
        e = eng.Engine(eng.PAGED_GEMM, B // 2, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"],
                       model["wv"], n_blocks=(B // 2) * 10)
        e.use_private_stream()
        for item_id, toks in items[r::2]:
            e.add_item(item_id, toks)
        engines.append(e)

    def drive(r):
        try:
            st = engines[r].run()
            assert st.finished == len(items[r::2]) and st.total_tokens > 0
            outs[r] = dict(engines[r].finished())
        except Exception as ex:  # surfaced below: an exception in a thread would otherwise pass silently
            errs.append(ex)

    threads = [threading.Thread(target=drive, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    got = {**outs[0], **outs[1]}
    for item_id, _ in items:
        assert len(got[item_id]) == len(cpu[item_id]) and (got[item_id] == cpu[item_id]).all(), item_id
    for e in engines:
        e.close()


def test_two_engines_with_different_compositions_in_one_process(oracle, mli, dev):
    """the lean / reference-launch-sequence switch used to be one process-wide flag, so an engine could
    change what another engine ran mid-flight.  Each engine now carries its own (mli_engine_configure) and installs it for
    the calling thread only: one engine on the reference's launch sequence and one on the lean compositions, driven by
    two threads at once, both give the CPU engine's tokens; so does a third engine created while the process default is
    'reference sequence' and flipped back afterwards."""
    import threading
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 160, 128, 1024
    model = make_model(59, V, S, D)
    items = make_items(60, 48, 1, 60)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    engines, outs, errs = [], [None, None, None], []
    for r, lean in enumerate((0, 1)):
        e = eng.Engine(eng.PAGED_GEMM, B // 2, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"],
                       model["wv"], n_blocks=(B // 2) * 5)
        e.configure(lean_layers=lean)
        e.use_private_stream()
        for item_id, toks in items[r::3]:
            e.add_item(item_id, toks)
        engines.append(e)
    mli.mli_engine_set_lean_layers(0)      # the default of engines created from here on ...
    e = eng.Engine(eng.PAGED, B // 2, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                   n_blocks=(B // 2) * 5)
    mli.mli_engine_set_lean_layers(1)      # ... which must not reach the engines that already exist
    e.use_private_stream()
    for item_id, toks in items[2::3]:
        e.add_item(item_id, toks)
    engines.append(e)

    def drive(r):
        try:
            st = engines[r].run()
            assert st.finished == len(items[r::3])
            outs[r] = dict(engines[r].finished())
        except Exception as ex:
            errs.append(ex)

    threads = [threading.Thread(target=drive, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    got = {**outs[0], **outs[1], **outs[2]}
    for item_id, _ in items:
        assert len(got[item_id]) == len(cpu[item_id]) and (got[item_id] == cpu[item_id]).all(), item_id
    for e in engines:
        e.close()


@pytest.mark.parametrize("kind_name,n_blocks_per_slot", [("PAGED", 4), ("PAGED_GEMM", 4), ("PAGED_GEMM", 8), ("PAGED", 5)])
def test_pipelined_engine_same_tokens_as_cpu_engine(oracle, mli, dev, kind_name, n_blocks_per_slot):
    """The pipelined loop (host one step behind the GPU, per-slot device updates, one-step-late refill, in-flight
    token of a preempted row dropped and regenerated) must leave every item's token stream unchanged; the tight pools
    force page growth and preemption while tokens are in flight."""
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 16, 128, 64, 1024
    model = make_model(53, V, S, D)
    items = make_items(54, 70, 1, 60)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    st, got = _run(getattr(eng, kind_name), model, items, B, S, n_blocks=n_blocks_per_slot * B, pipelined=True)
    assert st.finished == len(items) and st.waiting == 0 and st.in_flight == 0
    for item_id, _ in items:
        assert len(got[item_id]) == len(cpu[item_id]) and (got[item_id] == cpu[item_id]).all(), item_id
    assert st.total_tokens == sum(len(cpu[i]) - len(t) for i, t in items)   # a dropped in-flight token is appended once, after regeneration


def test_pipelined_engine_large_batch_with_preemption(mli, dev):
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 64, 160, 256, 1024
    model = make_model(55, V, S, D)
    items = make_items(56, 3 * B, 1, 63)
    _, seq = _run(eng.PAGED_GEMM, model, items, B, S, n_blocks=4 * B)
    st, pip = _run(eng.PAGED_GEMM, model, items, B, S, n_blocks=4 * B, pipelined=True)
    assert st.finished == len(items)
    for item_id, _ in items:
        assert len(pip[item_id]) == len(seq[item_id]) and (pip[item_id] == seq[item_id]).all(), item_id


@pytest.mark.parametrize("seed", range(10))
def test_pipelined_engine_random_configurations(mli, dev, seed):
    """Randomised shapes and pool sizes (down to pools that can hold only a few rows at once: constant preemption):
    the pipelined loop and the sequential loop must agree item for item, and both must drain the queue."""
    from min_llm_inference_amd import engine as eng
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(2, 40))
    S = 16 * int(rng.integers(2, 11))
    D = 4 * int(rng.integers(4, 40))
    V = 1024 + int(rng.integers(0, 300))
    n_items = int(rng.integers(1, 3 * B + 2))
    max_prompt = max(1, min(S - 2, int(rng.integers(1, S))))
    width = S // 16
    n_blocks = max(width + 4, int(rng.integers(2, 8)) * B)       # always enough for one full-length row
    kind = [eng.PAGED, eng.PAGED_GEMM][seed % 2]
    model = make_model(2000 + seed, V, S, D)
    items = make_items(3000 + seed, n_items, 1, max_prompt)
    st_seq, seq = _run(kind, model, items, B, S, n_blocks=n_blocks)
    st_pip, pip = _run(kind, model, items, B, S, n_blocks=n_blocks, pipelined=True)
    assert st_seq.finished == n_items and st_pip.finished == n_items, (B, S, D, n_items, n_blocks)
    assert st_seq.total_tokens == st_pip.total_tokens
    for item_id, _ in items:
        assert len(pip[item_id]) == len(seq[item_id]) and (pip[item_id] == seq[item_id]).all(), (item_id, B, S, D, n_blocks)


@pytest.mark.parametrize("seed", range(8))
def test_pipelined_engine_multi_round_random_configurations(oracle, mli, dev, seed):
    """n_forward_rounds > 1 in the pipelined loop (up to R tokens of a row in flight, rows finishing in the middle of a
    forward, preemption dropping up to R generated tokens): same tokens per item as the sequential multi-round loop,
    as the one-round engine, and as the CPU engine."""
    from min_llm_inference_amd import engine as eng
    from engine_sim import run_cpu_engine
    rng = np.random.default_rng(5000 + seed)
    B = int(rng.integers(2, 32))
    S = 16 * int(rng.integers(3, 11))
    D = 4 * int(rng.integers(4, 40))
    V = 1024 + int(rng.integers(0, 300))
    rounds = int(rng.integers(2, 9))
    n_items = int(rng.integers(1, 3 * B + 2))
    max_prompt = max(1, min(S - 2 - rounds, int(rng.integers(1, S))))
    n_blocks = max(S // 16 + 4, int(rng.integers(2, 8)) * B)
    kind = [eng.PAGED, eng.PAGED_GEMM][seed % 2]
    model = make_model(6000 + seed, V, S, D)
    items = make_items(7000 + seed, n_items, 1, max_prompt)
    st_seq, seq = _run(kind, model, items, B, S, n_blocks=n_blocks, rounds=rounds)
    st_pip, pip = _run(kind, model, items, B, S, n_blocks=n_blocks, rounds=rounds, pipelined=True)
    st_one, one = _run(kind, model, items, B, S, n_blocks=n_blocks, rounds=1, pipelined=True)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    assert st_seq.finished == n_items and st_pip.finished == n_items, (B, S, D, n_items, n_blocks, rounds)
    for item_id, _ in items:
        for name, got in (("sequential", seq), ("pipelined", pip), ("one round", one)):
            assert len(got[item_id]) == len(cpu[item_id]) and (got[item_id] == cpu[item_id]).all(), \
                (name, item_id, B, S, D, n_blocks, rounds)


@pytest.mark.parametrize("seed", range(8))
def test_pipelined_engine_pools_smaller_than_the_table_width(mli, dev, seed):
    """n_blocks in [4, width) and several forward rounds.  The pipelined loop asks for tokens + 2 R positions
    before result(step) is known; when the only row in flight cannot get that look-ahead page it must not preempt itself
    (the row may be emitting EOF in the forward that is still running) but take the reference's order for that
    iteration.  Whatever the sequential loop does with the workload -- finish it, or report the pool as too small -- the
    pipelined loop must do too, with the same tokens.  (EOF row scaled x3: rows end on EOF after ~10 tokens.)"""
    from min_llm_inference_amd import MliError, engine as eng
    rng = np.random.default_rng(8000 + seed)
    B = int(rng.integers(1, 5))
    S = 16 * int(rng.integers(6, 13))
    width = S // 16
    n_blocks = int(rng.integers(4, width))
    rounds = [1, 2, 3, 4, 6, 8, 2, 5][seed]
    D, V = 64, 1024
    model = make_model(8100 + seed, V, S, D)
    model["emb_table"][1023] *= 3.0
    items = make_items(8200 + seed, int(rng.integers(1, 9)), 1, max(1, min(n_blocks * 16 - 2 * rounds - 2, 40)))

    def run(pipelined):
        try:
            return _run(eng.PAGED, model, items, B, S, n_blocks=n_blocks, rounds=rounds, pipelined=pipelined)
        except MliError as e:
            assert "too small" in str(e)
            return None, None

    st_seq, seq = run(False)
    st_pip, pip = run(True)
    assert (st_seq is None) == (st_pip is None), (B, S, n_blocks, rounds, "sequential " + ("reported" if st_seq is None else "finished"))
    if st_seq is not None:
        assert st_seq.finished == len(items) and st_pip.finished == len(items)
        for item_id, _ in items:
            assert len(pip[item_id]) == len(seq[item_id]) and (pip[item_id] == seq[item_id]).all(), (item_id, B, S, n_blocks, rounds)


@pytest.mark.parametrize("kind_name,rounds,pipelined", [("PAGED", 1, True), ("PAGED_GEMM", 3, True), ("PAGED", 2, False)])
def test_step_graph_replay_gives_the_same_tokens(oracle, mli, dev, kind_name, rounds, pipelined):
    """Decode forwards replayed from a hipGraph (recorded on the engine's private stream at the second pure decode
    forward) == the same forwards launched one kernel at a time: tokens per item equal the CPU engine's, with
    preemption and prefill forwards (eager) in between the replays."""
    from min_llm_inference_amd import engine as eng
    from engine_sim import run_cpu_engine
    B, S, D, V = 24, 160, 128, 1024
    model = make_model(95, V, S, D)
    items = make_items(96, 3 * B, 1, 50)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    mli.mli_engine_set_step_graphs(1)
    try:
        e = eng.Engine(getattr(eng, kind_name), B, S, D, V, model["emb_table"], model["pos_table"], model["wk"],
                       model["wq"], model["wv"], n_blocks=4 * B, n_forward_rounds=rounds)
        e.use_private_stream()
        e.set_pipelined(pipelined)
        for item_id, toks in items:
            e.add_item(item_id, toks)
        st = e.run()
        got = {i: t for i, t in e.finished()}
        e.close()
    finally:
        mli.mli_engine_set_step_graphs(0)
    assert st.finished == len(items)
    for item_id, _ in items:
        assert len(got[item_id]) == len(cpu[item_id]) and (got[item_id] == cpu[item_id]).all(), item_id


def test_default_engine_loop_is_the_pipelined_one_where_it_applies(mli, dev):
    """mli_engine_run without mli_engine_set_pipelined: the pipelined loop for a paged engine (fewer iterations are not
    a criterion -- the same number of forwards --, so the check is that stepping is still possible only on an engine
    that was never run, and that tokens equal the explicit sequential run); the quirk keeps the sequential loop."""
    from min_llm_inference_amd import engine as eng
    B, S, D, V = 24, 128, 64, 1024
    model = make_model(91, V, S, D)
    items = make_items(92, 2 * B + 3, 1, 40)
    _, seq = _run(eng.PAGED, model, items, B, S, n_blocks=4 * B, pipelined=False)
    e = eng.Engine(eng.PAGED, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                   n_blocks=4 * B)
    for item_id, toks in items:
        e.add_item(item_id, toks)
    st = e.run()
    got = {i: t for i, t in e.finished()}
    e.close()
    assert st.finished == len(items)
    for item_id, _ in items:
        assert (got[item_id] == seq[item_id]).all()
    st_q, _ = _run(eng.PAGED, model, items, B, S, n_blocks=4 * B, quirk=True)   # _run selects sequential explicitly
    assert st_q.finished == len(items)


@pytest.mark.parametrize("pipelined", [False, True])
def test_pool_that_cannot_hold_a_rows_growth_is_an_error_not_a_hang(mli, dev, pipelined):
    """a pool that admits a row (>= 4 pages) but cannot hold its growth, no EOF.  The sequential loop ends in
    'pool too small'; the pipelined loop used to preempt / re-admit the row forever."""
    from min_llm_inference_amd import MliError, engine as eng
    B, S, D, V = 2, 160, 64, 1024
    model = make_model(58, V, S, D)
    model["emb_table"][1023] = 0.0   # EOF never wins the argmax
    e = eng.Engine(eng.PAGED, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                   n_blocks=5)
    e.set_pipelined(pipelined)
    e.add_item(0, np.array([1, 2, 3], np.int32))
    with pytest.raises(MliError, match="too small"):
        e.run()
    e.close()


@pytest.mark.parametrize("pipelined", [False, True])
def test_pool_too_small_for_any_row_is_an_error_not_a_hang(mli, dev, pipelined):
    """Fewer pages than a row's initial allotment: the reference's loop would spin forever; here run() fails."""
    from min_llm_inference_amd import MliError, engine as eng
    B, S, D, V = 4, 64, 64, 1024
    model = make_model(57, V, S, D)
    e = eng.Engine(eng.PAGED, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"],
                   n_blocks=3)
    if pipelined:
        e.set_pipelined()
    e.add_item(0, np.arange(5, dtype=np.int32))
    with pytest.raises(MliError, match="page pool is too small"):
        e.run()
    e.close()
