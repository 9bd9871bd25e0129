"""CPU model of the reference's contiguous engine (BASELINE config 1, 'CPU reference path'): the oracle's
host functions composed in InferenceModel::forward order (src/inference_model.cpp:14-39) and driven by the
scheduling logic of start_inference_engine (src/inferencer.cpp:11-41, src/item_storage.cpp:97-180).
TEST INFRASTRUCTURE: used to check the GPU engines' token streams, never by the product.

Greedy decoding is per row, so an item's output depends only on its own prompt: engines that schedule
differently (contiguous, paged, paged with preemption) must still produce the same tokens per item.
"""
import numpy as np


def make_model(seed, n_vocab, n_sequence, emb_dim):
    rng = np.random.default_rng(seed)

    def u(*shape, scale=1.0):
        return ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(scale)).astype(np.float32)

    sc = 2.0 / np.sqrt(emb_dim)
    return {"emb_table": u(n_vocab, emb_dim), "pos_table": u(n_sequence, emb_dim, scale=0.5),
            "wk": u(emb_dim, emb_dim, scale=sc), "wq": u(emb_dim, emb_dim, scale=sc), "wv": u(emb_dim, emb_dim, scale=sc)}


def make_items(seed, n_items, min_len, max_len, eof_token=1023):
    rng = np.random.default_rng(seed)
    return [(i, rng.integers(0, eof_token, size=int(rng.integers(min_len, max_len + 1))).astype(np.int32))
            for i in range(n_items)]


def run_cpu_engine(oracle, model, items, n_batch, n_sequence):
    """Returns {item id: all tokens (prompt + generated)} and the number of iterations."""
    D = model["wk"].shape[0]
    V = model["emb_table"].shape[0]
    B, S = n_batch, n_sequence
    inp = np.zeros((B, S), np.int32)
    lengths = np.zeros((B,), np.int32)
    inp_emb = np.zeros((B, S, D), np.float32)
    kt = np.zeros((B, D, S), np.float32)
    v = np.zeros((B, S, D), np.float32)
    q = np.zeros((B, D), np.float32)
    qkt = np.zeros((B, S), np.float32)
    att = np.zeros((B, D), np.float32)
    score = np.zeros((B, V), np.float32)
    result = np.zeros((B,), np.int32)
    new_idx = np.zeros((B,), np.int32)
    queue = [(i, list(map(int, t))) for i, t in items]
    processing, finished = {}, {}

    def insert(free_slots):
        n_new = 0
        for k, slot in enumerate(free_slots):
            new_idx[k] = slot
            if queue:
                item_id, toks = queue.pop(0)
                lengths[slot] = len(toks)
                inp[slot, :len(toks)] = toks
                processing[slot] = (item_id, toks)
                n_new += 1
            else:
                lengths[slot] = 0
        return n_new

    n_new = insert(list(range(B)))
    iterations = 0
    while processing or queue:
        oracle.inference_optimized_encoder_host(model["emb_table"], model["pos_table"], inp, inp_emb, lengths, new_idx, n_new)
        oracle.self_attention_inference_host(inp_emb, lengths, model["wk"], model["wq"], model["wv"], new_idx, kt, v, q,
                                             qkt, att, n_new)
        oracle.decoder_host(att, model["emb_table"], score, model["pos_table"], inp_emb, lengths, result)
        free_slots = []
        for b in range(B):
            tok = int(result[b])
            if tok == oracle.EMPTY_ROW_TOKEN_ID:
                free_slots.append(b)
                continue
            item_id, toks = processing[b]
            toks.append(tok)
            if len(toks) >= S or tok == oracle.EOF_TOKEN_ID:
                finished[item_id] = np.asarray(toks, np.int32)
                del processing[b]
                free_slots.append(b)
        n_new = insert(free_slots)
        iterations += 1
    return finished, iterations
