"""ctypes front end of the engine C ABI (include/mli_engine.h): continuous batching on one GPU.

Mirrors how the reference's drivers use the engine (tests/paged_for_profile.cpp:10-62): build the item queue
and the model, then run start_*_engine to completion.  Nothing is computed in Python.
"""
import ctypes

import numpy as np

from ._lib import EngineConfig, EngineStats, MliError, ShardStats, load_library

CONTIGUOUS, PAGED, PAGED_GEMM, PAGED_BF16 = 0, 1, 2, 3  # PAGED_BF16: extension, bf16 pages and weights
PAGED_FP8 = 4  # extension, opt-in: fp8 (OCP e4m3) pages, bf16 weights


def _fp(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    def __init__(self, kind, n_batch, n_sequence, emb_dim, n_vocab, emb_table, pos_table, wk, wq, wv, n_blocks=0,
                 n_forward_rounds=1, device=0, reference_length_reset_quirk=False):
        self._lib = load_library()
        self.cfg = EngineConfig(kind, n_batch, n_sequence, emb_dim, n_vocab, n_blocks, n_forward_rounds, device,
                                int(reference_length_reset_quirk))
        keep = [_fp(x) for x in (emb_table, pos_table, wk, wq, wv)]
        assert keep[0][0].shape == (n_vocab, emb_dim) and keep[1][0].shape == (n_sequence, emb_dim)
        self._h = ctypes.c_void_p()
        self._check(self._lib.mli_engine_create(ctypes.byref(self.cfg), *[k[1] for k in keep], ctypes.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise MliError("Hip Failure: " + (self._lib.mli_engine_last_error() or b"").decode())

    def use_private_stream(self):
        self._check(self._lib.mli_engine_use_private_stream(self._h))

    def configure(self, lean_layers=None, step_graphs=None):
        """This engine's own composition / replay switches (mli_engine_configure); None leaves a value as it is."""
        self._check(self._lib.mli_engine_configure(self._h, -1 if lean_layers is None else int(lean_layers),
                                                   -1 if step_graphs is None else int(step_graphs)))

    def set_pipelined(self, enabled=True):
        self._check(self._lib.mli_engine_set_pipelined(self._h, int(enabled)))

    def add_item(self, item_id, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        self._check(self._lib.mli_engine_add_item(self._h, int(item_id), t.ctypes.data_as(ctypes.c_void_p), len(t)))

    def run(self):
        st = EngineStats()
        self._check(self._lib.mli_engine_run(self._h, ctypes.byref(st)))
        return st

    def step(self):
        done = ctypes.c_int(0)
        self._check(self._lib.mli_engine_step(self._h, ctypes.byref(done)))
        return bool(done.value)

    def stats(self):
        st = EngineStats()
        self._check(self._lib.mli_engine_get_stats(self._h, ctypes.byref(st)))
        return st

    def decoder_result_ptr(self):
        p = ctypes.c_void_p()
        n = ctypes.c_int()
        self._check(self._lib.mli_engine_decoder_result(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def finished(self):
        """[(id, tokens)] in completion order."""
        out = []
        cap = self.cfg.n_sequence + 16
        buf = np.empty(cap, np.int32)
        for i in range(self.stats().finished):
            item_id = ctypes.c_int()
            n = ctypes.c_int()
            self._check(self._lib.mli_engine_get_finished(self._h, i, ctypes.byref(item_id),
                                                          buf.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n)))
            out.append((item_id.value, buf[:n.value].copy()))
        return out

    def close(self):
        if self._h:
            self._lib.mli_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedEngine(Engine):
    """An engine owned by a ShardGroup: same accessors, never destroyed from here."""

    def __init__(self, lib, handle, cfg):
        self._lib, self._h, self.cfg = lib, ctypes.c_void_p(handle), cfg

    def close(self):
        self._h = ctypes.c_void_p()


class ShardGroup:
    """include/mli_shard.h: one engine per GPU inside this process (one host thread each), stepped in lock step with one RCCL
    all-gather of the generated token ids per rank and iteration.  `n_batch` / `n_blocks` are PER RANK."""

    def __init__(self, kind, n_batch, n_sequence, emb_dim, n_vocab, emb_table, pos_table, wk, wq, wv, devices, n_blocks=0,
                 n_forward_rounds=1, loopback_ranks=0):
        """devices: one ordinal per rank (RCCL communicator); or loopback_ranks = N: N ranks on devices[0], exchange by
        device-to-device copies (mli_shard_group_create_loopback)."""
        self._lib = load_library()
        self.devices = [int(d) for d in devices]
        self.cfg = EngineConfig(kind, n_batch, n_sequence, emb_dim, n_vocab, n_blocks, n_forward_rounds, 0, 0)
        keep = [_fp(x) for x in (emb_table, pos_table, wk, wq, wv)]
        dev = np.asarray(self.devices, dtype=np.int32)
        self._h = ctypes.c_void_p()
        if loopback_ranks:
            self._check(self._lib.mli_shard_group_create_loopback(ctypes.byref(self.cfg), int(loopback_ranks), self.devices[0],
                                                                  *[k[1] for k in keep], ctypes.byref(self._h)))
            self.devices = [self.devices[0]] * int(loopback_ranks)
            return
        self._check(self._lib.mli_shard_group_create(ctypes.byref(self.cfg), len(self.devices),
                                                     dev.ctypes.data_as(ctypes.c_void_p), *[k[1] for k in keep],
                                                     ctypes.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise MliError("Hip Failure: " + (self._lib.mli_shard_last_error() or b"").decode())

    def add_item(self, item_id, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        self._check(self._lib.mli_shard_group_add_item(self._h, int(item_id), t.ctypes.data_as(ctypes.c_void_p), len(t)))

    def run(self):
        st = ShardStats()
        self._check(self._lib.mli_shard_group_run(self._h, ctypes.byref(st)))
        return st

    def gathered_ptr(self, rank):
        p, n = ctypes.c_void_p(), ctypes.c_int()
        self._check(self._lib.mli_shard_group_gathered(self._h, rank, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def engine(self, rank):
        h = self._lib.mli_shard_group_engine(self._h, rank)
        if not h:
            raise MliError("no such rank")
        return _BorrowedEngine(self._lib, h, self.cfg)

    def close(self):
        if self._h:
            self._lib.mli_shard_group_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
