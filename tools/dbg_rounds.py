import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from engine_sim import make_items, make_model, run_cpu_engine
from min_llm_inference_amd import engine as eng
S, D, V = 128, 64, 1024
model = make_model(45, V, S, D)
def run(kind, B, items, n_blocks, rounds):
    e = eng.Engine(kind, B, S, D, V, model["emb_table"], model["pos_table"], model["wk"], model["wq"], model["wv"], n_blocks=n_blocks, n_forward_rounds=rounds)
    for i, t in items: e.add_item(i, t)
    st = e.run(); out = dict(e.finished()); e.close(); return out
for B, n_items in ((1, 1), (1, 3), (16, 16), (16, 17), (16, 40)):
    items = make_items(46, n_items, 1, 60)
    cpu, _ = run_cpu_engine(oracle, model, items, B, S)
    for rounds in (1, 2):
        got = run(eng.PAGED, B, items, 8 * B, rounds)
        bad = []
        for i, t in items:
            a, b = got[i], cpu[i]
            d = np.nonzero(a != b)[0]
            if len(d): bad.append((i, len(t), int(d[0])))
        print(f"B {B} items {n_items} rounds {rounds}: mismatches (id, prompt_len, first_diff) {bad}")
