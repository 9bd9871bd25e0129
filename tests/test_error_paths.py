"""Error convention of the C ABI (include/mli_kernels.h: 0, hipError_t > 0, MLI_ERR_* < 0), checked where the
library rejects a call BEFORE touching the GPU -- so these run in the CPU suite.  The reference asserts on the same
preconditions (src/kernels/paged_attention.cu:105-107: n_sequence % PAGE_BLOCK_SIZE, emb_dim % 4) or returns early
(n_new_items == 0: self_attention_inference_optimized.cu:308-310, paged_attention.cu:100-102)."""
import ctypes

BAD_ARG, WORKSPACE = -22, -12
NULL = ctypes.c_void_p(0)


def test_paged_entry_points_reject_bad_shapes(mli):
    # n_sequence not a multiple of the page size
    assert mli.mli_get_latest_k_q_v_paged(NULL, NULL, NULL, NULL, NULL, NULL, 4, 100, 64, NULL) == BAD_ARG
    assert mli.mli_fill_new_k_v_cache_paged(NULL, NULL, NULL, NULL, NULL, 4, 100, 64, 2, NULL) == BAD_ARG
    # emb_dim not a multiple of 4 (fp32) / 8 (bf16)
    assert mli.mli_get_latest_k_q_v_paged(NULL, NULL, NULL, NULL, NULL, NULL, 4, 128, 66, NULL) == BAD_ARG
    assert mli.mli_get_latest_k_q_v_paged_bf16(NULL, NULL, NULL, NULL, NULL, NULL, 4, 128, 68, NULL) == BAD_ARG
    # empty batch
    assert mli.mli_get_latest_k_q_v_paged(NULL, NULL, NULL, NULL, NULL, NULL, 0, 128, 64, NULL) == BAD_ARG
    assert mli.mli_get_latest_kt_q_v(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, 128, 64, 64, NULL) == BAD_ARG
    # negative n_new_items
    assert mli.mli_fill_new_k_v_cache_paged(NULL, NULL, NULL, NULL, NULL, 4, 128, 64, -1, NULL) == BAD_ARG


def test_no_new_items_is_a_no_op(mli):
    """Nothing is launched and no pointer is read (they are all NULL here)."""
    assert mli.mli_fill_new_kt_v_cache(NULL, NULL, NULL, NULL, NULL, NULL, NULL, 4, 128, 64, 64, 0, NULL) == 0
    assert mli.mli_fill_new_k_v_cache_paged(NULL, NULL, NULL, NULL, NULL, 4, 128, 64, 0, NULL) == 0
    assert mli.mli_fill_new_k_v_cache_paged_bf16(NULL, NULL, NULL, NULL, NULL, 4, 128, 64, 0, NULL) == 0


def test_decode_scan_argument_checks(mli):
    assert mli.mli_decode_scan_paged(NULL, NULL, NULL, NULL, NULL, 4, 128, 64, 0, 0, NULL, 0, NULL) == BAD_ARG   # phases
    assert mli.mli_decode_scan_paged(NULL, NULL, NULL, NULL, NULL, 4, 128, 64, 0, 4, NULL, 0, NULL) == BAD_ARG
    # emb_dim beyond what the single-pass kernel covers, and a multi-chunk problem without workspace
    assert mli.mli_decode_scan_paged(NULL, NULL, NULL, NULL, NULL, 4, 128, 4096, 0, 3, NULL, 0, NULL) == BAD_ARG
    assert mli.mli_decode_scan_paged(NULL, NULL, NULL, NULL, NULL, 1024, 4096, 512, 1, 3, NULL, 0, NULL) == BAD_ARG


def test_softmax_v_needs_its_workspace(mli):
    need = mli.mli_attention_workspace_bytes(1024, 4096, 512)
    assert need > 0
    assert mli.mli_softmax_v_paged(NULL, NULL, NULL, NULL, 1024, 4096, 512, NULL, 0, NULL) == WORKSPACE
    assert mli.mli_softmax_v_paged(NULL, NULL, NULL, NULL, 1024, 4096, 512, NULL, need - 1, NULL) == WORKSPACE


def test_tune_rejects_unknown_keys_and_values(mli):
    assert mli.mli_tune(b"no_such_knob", 1) == BAD_ARG
    assert mli.mli_tune(b"chunk_tokens", 100) == BAD_ARG      # not a power of two
    assert mli.mli_tune(b"chunk_tokens", 4096) == BAD_ARG     # beyond the largest chunk
    assert mli.mli_tune(b"qkt_token_batch", 5) == BAD_ARG
    assert mli.mli_tune(b"chunk_tokens", 0) == 0


def test_engine_rejects_bad_configurations(mli):
    from min_llm_inference_amd._lib import EngineConfig
    import numpy as np
    w = np.zeros((16, 16), np.float32)
    p = w.ctypes.data_as(ctypes.c_void_p)
    h = ctypes.c_void_p()
    bad = [
        EngineConfig(7, 4, 128, 16, 1100, 16, 1, 0, 0),     # unknown kind
        EngineConfig(1, 4, 100, 16, 1100, 16, 1, 0, 0),     # paged, n_sequence % 16
        EngineConfig(1, 4, 128, 18, 1100, 16, 1, 0, 0),     # emb_dim % 4
        EngineConfig(1, 4, 128, 16, 1000, 16, 1, 0, 0),     # vocabulary without the EOF token id
        EngineConfig(1, 4, 128, 16, 1100, 0, 1, 0, 0),      # no pages
        EngineConfig(1, 4, 128, 16, 1100, 16, 17, 0, 0),    # more rounds than a page has slots
        EngineConfig(3, 4, 128, 20, 1100, 16, 1, 0, 0),     # bf16: emb_dim % 8
    ]
    for cfg in bad:
        assert mli.mli_engine_create(ctypes.byref(cfg), p, p, p, p, p, ctypes.byref(h)) == -1
        assert b"invalid engine configuration" in mli.mli_engine_last_error()
    ok = EngineConfig(1, 4, 128, 16, 1100, 16, 1, 0, 0)
    assert mli.mli_engine_create(ctypes.byref(ok), None, p, p, p, p, ctypes.byref(h)) == -1
    assert b"null argument" in mli.mli_engine_last_error()
