"""The decode step of a small paged batch as ONE launch (decode_step_fused.hip, opt-in through mli_tune "step_fused" = 1:
projection tiles, scan items, logits tiles and the token pick as roles of one grid, rows handed over through counters in
the workspace) against the same step as separate launches (the default): everything a caller can observe -- tokens of every step, lengths,
q_output, attention_result and every byte of the page pool -- must be IDENTICAL, over several steps (the counters
must be back at zero after each launch), with empty rows, rows that finish (EOF, sequence full), ragged batch / vocab
sizes, both row widths of the scan and under graph replay; and the launch's error word must stay zero.
The separate launches are themselves checked against the oracle in test_lean_path_gpu.py / test_paged_kernels_gpu.py;
the full-size config-3 case here is checked against a float64 model of the step as well."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import PAGE, assert_close, assert_equal

pytestmark = pytest.mark.gpu

EOF = 1023


class State:
    """Device state of one paged decode batch: every row owns all its S/16 pages (shuffled pool), contents U(-1, 1)."""

    def __init__(self, dev, seed, B, S, D, V, lengths, n_results=1):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        rng = np.random.default_rng(seed)
        u = lambda *shape: torch.rand(*shape, device=dev, generator=g) * 2 - 1
        self.B, self.S, self.D, self.V, self.n_results = B, S, D, V, n_results
        W = S // PAGE
        block = PAGE * 3 * D
        self.pool = u(B * W * block)
        order = rng.permutation(B * W).reshape(B, W)
        self.page_table = torch.from_numpy(self.pool.data_ptr() + order.astype(np.int64) * block * 4).to(dev)
        self.lengths = torch.from_numpy(np.asarray(lengths, np.int32)).to(dev)
        sc = 1.0 / np.sqrt(D)
        self.wk, self.wq, self.wv = (u(D, D) * sc for _ in range(3))
        self.emb_table = u(V, D)
        self.wpe = u(S, D)
        self.q_output = torch.full((B, D), 7.0, device=dev)
        self.attention_result = torch.full((B, D), 7.0, device=dev)
        self.decoder_result = torch.full((B, n_results), -7, dtype=torch.int32, device=dev)

    def clone(self):
        c = object.__new__(State)
        c.__dict__.update(self.__dict__)
        for k in ("pool", "lengths", "q_output", "attention_result", "decoder_result"):
            setattr(c, k, getattr(self, k).clone())
        base, new = self.pool.data_ptr(), c.pool.data_ptr()
        c.page_table = self.page_table - base + new
        return c

    def step(self, mli, ops, i_result=0):
        ws, need = ops.workspace_for(self.B, self.S, self.D, self.pool.device)
        sc, sc_need = ops.decoder_scratch_for(self.B, self.V, self.pool.device)
        self._keep = (ws, sc)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = mli.mli_paged_decode_step(p(self.page_table), p(self.lengths), p(self.wk), p(self.wq), p(self.wv),
                                       p(self.emb_table), p(self.wpe), p(self.q_output), p(self.attention_result),
                                       p(self.decoder_result), self.B, self.S, self.D, self.V, self.n_results, i_result, 0,
                                       p(ws), need, p(sc), sc_need,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
        return ws, need

    def observable(self):
        torch.cuda.synchronize()
        return {k: getattr(self, k).cpu().numpy() for k in ("pool", "lengths", "q_output", "attention_result", "decoder_result")}


def _error_word(mli, ws, need):
    code = ctypes.c_uint(99)
    assert mli.mli_debug_step_fused_error(ctypes.c_void_p(ws.data_ptr()), need, ctypes.byref(code)) == 0
    return code.value


def _lengths(rng, B, S, zero_every=None, full_rows=()):
    L = rng.integers(1, S - 1, size=B).astype(np.int32)
    if zero_every:
        L[::zero_every] = 0
    for b in full_rows:
        L[b] = S - 1          # the step's token is the row's last: it finishes (lengths -> 0)
    return L


def _run_both(mli, dev, st0, n_steps):
    from min_llm_inference_amd import ops
    out = []
    for fused in (1, 0):
        assert mli.mli_tune(b"step_fused", fused) == 0
        st = st0.clone()
        toks = []
        for i in range(n_steps):
            ws, need = st.step(mli, ops, i % st.n_results)
            toks.append(st.decoder_result.clone())
        obs = st.observable()
        obs["tokens"] = torch.stack(toks).cpu().numpy()
        if fused:
            assert _error_word(mli, ws, need) == 0
        out.append(obs)
    mli.mli_tune(b"step_fused", 0)
    return out


# (seed, B, S, D, V): config 3's shape; ragged batch and vocabulary; 512-wide rows (two loads per row); a single row block
SHAPES = [(501, 256, 1024, 256, 1024), (502, 75, 256, 128, 200), (503, 40, 512, 512, 96), (504, 9, 128, 64, 33),
          (505, 130, 2048, 64, 1024)]


@pytest.mark.parametrize("seed,B,S,D,V", SHAPES)
def test_one_launch_step_equals_the_separate_launches(mli, dev, seed, B, S, D, V):
    rng = np.random.default_rng(seed)
    L = _lengths(rng, B, S, zero_every=7, full_rows=(1, B - 1))
    st0 = State(dev, seed, B, S, D, V, L, n_results=3)
    fused, separate = _run_both(mli, dev, st0, n_steps=4)
    for k in ("tokens", "lengths", "decoder_result", "q_output", "attention_result", "pool"):
        assert_equal(fused[k], separate[k], what=k)
    # the step did something: non-empty rows got a token, empty rows the marker, full rows finished
    t0 = fused["tokens"][0][:, 0]
    assert (t0[L == 0] == -1).all() and (t0[L > 0] >= 0).all()
    assert fused["lengths"][1] == 0 and fused["lengths"][B - 1] == 0
    alive = (L > 0) & (L < S - 5)
    alive[[1, B - 1]] = False
    if V <= EOF:
        assert (fused["lengths"][alive] == L[alive] + 4).all()


def test_rows_that_emit_eof_finish(mli, dev):
    B, S, D, V = 96, 512, 128, 1024
    rng = np.random.default_rng(77)
    st0 = State(dev, 77, B, S, D, V, _lengths(rng, B, S))
    # make EOF the argmax of about half of the rows: its embedding = 4 x the attention output direction of those rows is
    # not known in advance, so give it a huge norm along a random direction -- rows whose result points that way pick it
    st0.emb_table[EOF] = st0.emb_table[EOF] * 50
    fused, separate = _run_both(mli, dev, st0, n_steps=3)
    for k in ("tokens", "lengths", "q_output", "attention_result", "pool"):
        assert_equal(fused[k], separate[k], what=k)
    n_eof = int((fused["tokens"][0][:, 0] == EOF).sum())
    assert 0 < n_eof < B, n_eof
    assert (fused["lengths"][fused["tokens"][0][:, 0] == EOF] == 0).all()


def test_one_launch_step_config3_against_a_float64_model(mli, dev):
    """BASELINE config 3 in full: attention_result and tokens of the one-launch step against a float64 numpy model of
    the step (the model tests/test_oracle.py holds the C oracle to)."""
    B, S, D, V = 256, 1024, 256, 1024
    rng = np.random.default_rng(9)
    L = rng.integers(S // 4, 3 * S // 4, size=B).astype(np.int32)
    st = State(dev, 9, B, S, D, V, L)
    from min_llm_inference_amd import ops
    pool0 = st.pool.cpu().numpy().copy()
    table = ((st.page_table.cpu().numpy() - st.pool.data_ptr()) // 4).astype(np.int64)
    mli.mli_tune(b"step_fused", 1)
    try:
        ws, need = st.step(mli, ops)
        obs = st.observable()
    finally:
        mli.mli_tune(b"step_fused", 0)
    assert _error_word(mli, ws, need) == 0
    W = S // PAGE
    blocks = pool0.reshape(-1, PAGE, 3, D)
    wk, wq, wv = (w.cpu().numpy() for w in (st.wk, st.wq, st.wv))
    emb = st.emb_table.cpu().numpy()
    want_attn = np.zeros((B, D), np.float32)
    for b in range(B):
        rows = blocks[table[b] // (PAGE * 3 * D)].reshape(W * PAGE, 3, D)
        x = rows[L[b] - 1, 0].astype(np.float64)
        k = rows[:L[b], 1].astype(np.float64).copy()
        v = rows[:L[b], 2].astype(np.float64).copy()
        k[L[b] - 1] = x @ wk.astype(np.float64)
        v[L[b] - 1] = x @ wv.astype(np.float64)
        q = x @ wq.astype(np.float64)
        s = k @ q / np.sqrt(D)
        p = np.exp(s - s.max())
        p /= p.sum()
        want_attn[b] = (p @ v).astype(np.float32)
    assert_close(obs["attention_result"], want_attn, thr=1e-3, what="attention_result")
    logits = want_attn.astype(np.float64) @ emb.astype(np.float64).T
    top2 = np.sort(logits, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-3      # rows whose argmax does not hang on rounding
    assert clear.sum() > B // 2
    assert_equal(obs["decoder_result"][clear, 0], logits.argmax(1)[clear].astype(np.int32), what="tokens")
    assert_equal(obs["lengths"], np.where(obs["decoder_result"][:, 0] == EOF, 0, L + 1).astype(np.int32), what="lengths")


def test_one_launch_step_graph_replay(mli, dev):
    """The one-launch step replayed from a hipGraph == launched eagerly (its counters reset themselves: no memset node)."""
    from min_llm_inference_amd import ops
    B, S, D, V = 128, 512, 256, 512
    rng = np.random.default_rng(31)
    st0 = State(dev, 31, B, S, D, V, _lengths(rng, B, S, zero_every=9))
    res = []
    mli.mli_tune(b"step_fused", 1)
    for graph in (False, True):
        st = st0.clone()
        side = torch.cuda.Stream(device=dev)
        toks = []
        with torch.cuda.stream(side):
            st.step(mli, ops)
            toks.append(st.decoder_result.clone())
            g = ops.StepGraph(lambda: st.step(mli, ops)) if graph else None
            for _ in range(5):
                g.launch() if graph else st.step(mli, ops)
                toks.append(st.decoder_result.clone())
            side.synchronize()
        obs = st.observable()
        obs["tokens"] = torch.stack(toks).cpu().numpy()
        res.append(obs)
    mli.mli_tune(b"step_fused", 0)
    for k in ("tokens", "lengths", "q_output", "attention_result", "pool"):
        assert_equal(res[1][k], res[0][k], what=k)
