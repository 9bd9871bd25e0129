"""BASELINE config 1 (batch=4, dim=64, max_seq=128) on the CPU path: the oracle's host functions composed in
forward order under the scheduler -- the reference's own pass criterion is finish_count == n_items
(tests/inferencer_test.cpp:43).  No GPU involved."""
import numpy as np

from engine_sim import make_items, make_model, run_cpu_engine


def test_config1_cpu_engine_finishes_every_item(oracle):
    B, D, S, V = 4, 64, 128, 1024
    model = make_model(31, V, S, D)
    items = make_items(32, 10, 1, 40)
    finished, iterations = run_cpu_engine(oracle, model, items, B, S)
    assert len(finished) == len(items)
    for item_id, prompt in items:
        toks = finished[item_id]
        assert (toks[:len(prompt)] == prompt).all()
        assert len(toks) == S or toks[-1] == oracle.EOF_TOKEN_ID
        assert len(toks) > len(prompt)
    assert iterations >= 1


def test_cpu_engine_is_scheduling_independent(oracle):
    """Per-item outputs do not depend on how many slots the batch has (rows are independent)."""
    D, S, V = 32, 64, 1024
    model = make_model(33, V, S, D)
    items = make_items(34, 7, 1, 20)
    a, _ = run_cpu_engine(oracle, model, items, 2, S)
    b, _ = run_cpu_engine(oracle, model, items, 7, S)
    for item_id, _ in items:
        assert (a[item_id] == b[item_id]).all()
