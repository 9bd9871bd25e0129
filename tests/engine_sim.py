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


class CpuEngine:
    """Steppable form: one call to step() = one iteration of start_inference_engine's loop."""

    def __init__(self, oracle, model, items, n_batch, n_sequence, bf16=False):
        """bf16 = True models the bf16 page extension: Wk/Wq/Wv and everything stored in a page (input embedding, K,
        V) are rounded to bfloat16 where the GPU path stores them; q, scores, sums and logits stay fp32.
        bf16 = "fp8" models the fp8 page extension: bf16 weights, page contents rounded to OCP e4m3 (saturating)."""
        self.o, self.m = oracle, dict(model)
        self.bf16 = bool(bf16)
        if bf16:
            from helpers import bf16_round, fp8_round
            self.round = fp8_round if bf16 == "fp8" else bf16_round
            for k in ("wk", "wq", "wv"):
                self.m[k] = bf16_round(model[k])
        D = model["wk"].shape[0]
        V = model["emb_table"].shape[0]
        B, S = n_batch, n_sequence
        self.B, self.S = B, S
        self.inp = np.zeros((B, S), np.int32)
        self.lengths = np.zeros((B,), np.int32)
        self.inp_emb = np.zeros((B, S, D), np.float32)
        self.kt = np.zeros((B, D, S), np.float32)
        self.v = np.zeros((B, S, D), np.float32)
        self.q = np.zeros((B, D), np.float32)
        self.qkt = np.zeros((B, S), np.float32)
        self.att = np.zeros((B, D), np.float32)
        self.score = np.zeros((B, V), np.float32)
        self.result = np.full((B,), -1, np.int32)
        self.new_idx = np.zeros((B,), np.int32)
        self.queue = [(i, list(map(int, t))) for i, t in items]
        self.processing, self.finished = {}, {}
        self.iterations = 0
        self.n_new = self._insert(list(range(B)))

    def _insert(self, free_slots):
        n_new = 0
        for k, slot in enumerate(free_slots):
            self.new_idx[k] = slot
            if self.queue:
                item_id, toks = self.queue.pop(0)
                self.lengths[slot] = len(toks)
                self.inp[slot, :len(toks)] = toks
                self.processing[slot] = (item_id, toks)
                n_new += 1
            else:
                self.lengths[slot] = 0
        return n_new

    def done(self):
        return not (self.processing or self.queue)

    def step(self):
        o, m = self.o, self.m
        o.inference_optimized_encoder_host(m["emb_table"], m["pos_table"], self.inp, self.inp_emb, self.lengths,
                                           self.new_idx, self.n_new)
        if not self.bf16:
            o.self_attention_inference_host(self.inp_emb, self.lengths, m["wk"], m["wq"], m["wv"], self.new_idx,
                                            self.kt, self.v, self.q, self.qkt, self.att, self.n_new)
        else:
            self.inp_emb[...] = self.round(self.inp_emb)
            o.fill_new_kt_v_cache(self.inp_emb, self.new_idx, self.lengths, m["wk"], m["wv"], self.kt, self.v,
                                  self.n_new)
            o.get_latest_kt_q_v(self.inp_emb, self.lengths, m["wk"], m["wq"], m["wv"], self.kt, self.v, self.q)
            self.kt[...] = self.round(self.kt)
            self.v[...] = self.round(self.v)
            o.qkt_host(self.q, self.kt, self.lengths, self.qkt)
            o.softmax_in_place_with_lengths_host(self.qkt, self.lengths)
            o.softmax_v_host(self.qkt, self.v, self.att, self.lengths)
        o.decoder_host(self.att, m["emb_table"], self.score, m["pos_table"], self.inp_emb, self.lengths, self.result)
        free_slots = []
        for b in range(self.B):
            tok = int(self.result[b])
            if tok == o.EMPTY_ROW_TOKEN_ID:
                free_slots.append(b)
                continue
            item_id, toks = self.processing[b]
            toks.append(tok)
            if len(toks) >= self.S or tok == o.EOF_TOKEN_ID:
                self.finished[item_id] = np.asarray(toks, np.int32)
                del self.processing[b]
                free_slots.append(b)
        self.n_new = self._insert(free_slots)
        self.iterations += 1
        return self.result.copy()


def run_cpu_engine(oracle, model, items, n_batch, n_sequence, bf16=False):
    """Returns {item id: all tokens (prompt + generated)} and the number of iterations."""
    e = CpuEngine(oracle, model, items, n_batch, n_sequence, bf16=bf16)
    while not e.done():
        e.step()
    return e.finished, e.iterations
