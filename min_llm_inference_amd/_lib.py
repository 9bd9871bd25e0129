"""ctypes loader for libmli_hip.so (C ABI: include/mli_kernels.h, include/mli_engine.h, include/mli_shard.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class MliError(RuntimeError):
    """Raised when a C-ABI call returns non-zero (the reference throws std::runtime_error("Cuda Failure"))."""


def library_path():
    return os.path.join(_HERE, "lib", "libmli_hip.so")


def load_library():
    """Load the HIP library.  There is deliberately no fallback: a missing .so is a hard error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own libamdhip64 / libhsa-runtime64.  Import it FIRST so that libmli_hip.so's
    # libamdhip64.so.7 dependency resolves to the copy already in the process; loading /opt/rocm's runtime
    # beside torch's gives two HSA runtimes in one process, and the second one to initialise sees no GPU.
    import torch  # noqa: F401
    path = library_path()
    if not os.path.exists(path):
        raise MliError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = ctypes.CDLL(path)
    _declare(lib)
    _LIB = lib
    return lib


_P = ctypes.c_void_p
_I = ctypes.c_int
_Z = ctypes.c_size_t

# name -> argtypes; restype is int unless listed in _RESTYPES.  Mirrors include/mli_kernels.h 1:1.
SIGNATURES = {
    "mli_abi_version": [],
    "mli_elem_supported": [_I],
    "mli_attention_workspace_bytes": [_I, _I, _I],
    "mli_attention_workspace_init": [_P, _Z, _P],
    "mli_fill_new_kt_v_cache": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "mli_get_latest_kt_q_v": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mli_qkt": [_P, _P, _P, _P, _I, _I, _I, _P],
    "mli_softmax_in_place_with_lengths": [_P, _P, _I, _I, _P],
    "mli_softmax_v": [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P],
    "mli_inference_self_attention": [_P] * 11 + [_I] * 5 + [_P, _Z, _P],
    "mli_fill_new_k_v_cache_paged": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mli_get_latest_k_q_v_paged": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "mli_qkt_paged": [_P, _P, _P, _P, _I, _I, _I, _P],
    "mli_softmax_v_paged": [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P],
    "mli_paged_attention": [_P] * 9 + [_I] * 4 + [_P, _Z, _P],
    "mli_fill_new_k_v_cache_paged_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "mli_get_latest_k_q_v_paged_bf16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "mli_qkt_paged_bf16": [_P, _P, _P, _P, _I, _I, _I, _P],
    "mli_softmax_v_paged_bf16": [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P],
    "mli_paged_attention_bf16": [_P] * 9 + [_I] * 4 + [_P, _Z, _P],
    "mli_paged_attention_encoder_bf16": [_P] * 6 + [_I] * 4 + [_P],
    "mli_paged_decoder_multi_rounds_bf16": [_P] * 7 + [_I] * 6 + [_P],
    "mli_decode_scan_paged": [_P] * 5 + [_I] * 5 + [_P, _Z, _P],
    "mli_paged_attention_lean": [_P] * 8 + [_I] * 5 + [_P, _Z, _P],
    "mli_get_latest_k_q_v_paged_lean": [_P] * 6 + [_I] * 4 + [_P],
    "mli_self_attention_lean": [_P] * 10 + [_I] * 5 + [_P, _Z, _P],
    "mli_decode_scan_contiguous": [_P] * 5 + [_I] * 3 + [_P, _Z, _P],
    "mli_decoder_scratch_bytes": [_I, _I],
    "mli_decoder_fused": [_P] * 6 + [_I] * 4 + [_P, _Z, _P],
    "mli_paged_decoder_fused": [_P] * 6 + [_I] * 7 + [_P, _Z, _P],
    "mli_paged_prefill": [_P] * 8 + [_I] * 5 + [_P],
    "mli_prefill": [_P] * 10 + [_I] * 5 + [_P],
    "mli_paged_decode_step": [_P] * 10 + [_I] * 7 + [_P, _Z, _P, _Z, _P],
    "mli_decode_step": [_P] * 13 + [_I] * 4 + [_P, _Z, _P, _Z, _P],
    "mli_graph_begin_capture": [_P],
    "mli_graph_end_capture": [_P, ctypes.POINTER(ctypes.c_void_p)],
    "mli_graph_launch": [_P, _P],
    "mli_stream_wait_stream": [_P, _P],
    "mli_graph_destroy": [_P],
    "mli_inference_optimized_encoder": [_P] * 6 + [_I] * 4 + [_P],
    "mli_paged_attention_encoder": [_P] * 6 + [_I] * 4 + [_P],
    "mli_decoder": [_P] * 7 + [_I] * 4 + [_P],
    "mli_paged_decoder_multi_rounds": [_P] * 7 + [_I] * 6 + [_P],
    "mli_clone_inp_embedding_k_v_cache": [_P] * 5 + [_I] * 3 + [_P],
    "mli_tune": [ctypes.c_char_p, _I],
    "mli_stream_copy": [_P, _P, _Z, _P],
    "mli_stream_read": [_P, _P, _Z, _P],
    "mli_f32_to_fp8": [_P, _P, _Z, _P],
}
class EngineConfig(ctypes.Structure):
    """mli_engine_config (include/mli_engine.h)."""
    _fields_ = [("kind", _I), ("n_batch", _I), ("n_sequence", _I), ("emb_dim", _I), ("n_vocab", _I),
                ("n_blocks", _I), ("n_forward_rounds", _I), ("device", _I), ("reference_length_reset_quirk", _I)]


class EngineStats(ctypes.Structure):
    """mli_engine_stats (include/mli_engine.h)."""
    _fields_ = [("total_tokens", ctypes.c_longlong), ("seconds", ctypes.c_double), ("iterations", ctypes.c_longlong),
                ("finished", _I), ("waiting", _I), ("in_flight", _I)]


_PP = ctypes.POINTER(ctypes.c_void_p)
_IP = ctypes.POINTER(_I)
ENGINE_SIGNATURES = {
    "mli_engine_create": [ctypes.POINTER(EngineConfig), _P, _P, _P, _P, _P, _PP],
    "mli_engine_destroy": [_P],
    "mli_engine_add_item": [_P, _I, _P, _I],
    "mli_engine_use_private_stream": [_P],
    "mli_engine_set_pipelined": [_P, _I],
    "mli_engine_run": [_P, ctypes.POINTER(EngineStats)],
    "mli_engine_step": [_P, _IP],
    "mli_engine_get_stats": [_P, ctypes.POINTER(EngineStats)],
    "mli_engine_decoder_result": [_P, _PP, _IP],
    "mli_engine_get_finished": [_P, _I, _IP, _P, _I, _IP],
    "mli_engine_configure": [_P, _I, _I],
    "mli_engine_set_lean_layers": [_I],
    "mli_engine_set_step_graphs": [_I],
    "mli_engine_last_error": [],
    "mli_engine_stream": [_P, _PP],
    # include/mli_shard.h: the row-sharded engine group (one engine per GPU, RCCL all-gather of the token ids)
    "mli_shard_group_create": [ctypes.POINTER(EngineConfig), _I, _P, _P, _P, _P, _P, _P, _PP],
    "mli_shard_group_create_loopback": [ctypes.POINTER(EngineConfig), _I, _I, _P, _P, _P, _P, _P, _PP],
    "mli_shard_group_destroy": [_P],
    "mli_shard_group_size": [_P],
    "mli_shard_group_add_item": [_P, _I, _P, _I],
    "mli_shard_group_run": [_P, _P],
    "mli_shard_group_gathered": [_P, _I, _PP, _IP],
    "mli_shard_group_engine": [_P, _I],
    "mli_shard_last_error": [],
}


class ShardStats(ctypes.Structure):
    """mli_shard_stats (include/mli_shard.h)."""
    _fields_ = [("total_tokens", ctypes.c_longlong), ("seconds", ctypes.c_double), ("iterations", ctypes.c_longlong),
                ("finished", _I), ("ranks_seen", _I), ("gather_us", ctypes.c_double)]


_RESTYPES = {"mli_shard_last_error": ctypes.c_char_p, "mli_shard_group_destroy": None, "mli_shard_group_engine": _P,
             "mli_attention_workspace_bytes": _Z, "mli_decoder_scratch_bytes": _Z, "mli_engine_last_error": ctypes.c_char_p,
             "mli_engine_destroy": None, "mli_engine_set_lean_layers": None,
             "mli_engine_set_step_graphs": None}


def _declare(lib):
    for name, argtypes in list(SIGNATURES.items()) + list(ENGINE_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _I)
