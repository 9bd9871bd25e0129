"""torch-tensor front end over the C ABI, one function per reference launcher (same names).

torch supplies device memory and the stream only; every computation happens in libmli_hip.so.
Argument order and meaning follow the reference headers
(include/kernels/self_attention_inference_optimized.h, include/kernels/paged_attention.h,
include/kernels/encoder.h, include/kernels/decoder.h); scalars the reference derives from
Tensor shapes are derived from the torch shapes here in the same way.
"""
import ctypes

import torch

from ._lib import MliError, load_library  # noqa: F401  (re-exported: ops.MliError, ops.load_library)

PAGE_BLOCK_SIZE = 16
EMPTY_ROW_TOKEN_ID = -1
EOF_TOKEN_ID = 1023

ELEM_F32, ELEM_BF16, ELEM_FP8 = 0, 1, 2  # MLI_ELEM_* of include/mli_kernels.h

_workspaces = {}


def has_fp8():
    """Whether this build of the library implements the fp8 (OCP e4m3) page extension."""
    return bool(load_library().mli_elem_supported(ELEM_FP8))


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise MliError("libmli_hip.so operates on device tensors only (no CPU fallback)")
    if not t.is_contiguous():
        raise MliError("tensors crossing the C ABI must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise MliError(f"Hip Failure: {what} returned {rc}")


def workspace_for(n_batch, n_sequence, dim, device):
    """Caller-owned scratch for the split-sequence kernels (grown on demand, never inside a timed region)."""
    lib = load_library()
    need = int(lib.mli_attention_workspace_bytes(n_batch, n_sequence, dim))
    # one buffer per (device, stream), as the C++ side keys its scratch (host/src/memory_hip.cpp): the split-sequence
    # kernels of two streams must not share partial sums.  A buffer that has to grow is replaced only after the
    # stream that used the old one has drained.
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(index)
    key = (index, stream.cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            stream.synchronize()
        ws = torch.zeros(max(need, 16), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws, need


# ---- contiguous ("naive") path ------------------------------------------------------------
def launch_fill_new_kt_v_cache(inp_embedding, new_batch_idx, lengths, wk, wv, kt_cache, v_cache, n_new_items):
    B, S, Din = inp_embedding.shape
    _check(load_library().mli_fill_new_kt_v_cache(_p(inp_embedding), _p(new_batch_idx), _p(lengths), _p(wk), _p(wv),
                                                  _p(kt_cache), _p(v_cache), B, S, Din, wk.shape[1], n_new_items,
                                                  _stream()), "mli_fill_new_kt_v_cache")


def launch_get_latest_kt_q_v(inp_embedding, lengths, wk, wq, wv, kt_cache, v_cache, q_output):
    B, S, Din = inp_embedding.shape
    _check(load_library().mli_get_latest_kt_q_v(_p(inp_embedding), _p(lengths), _p(wk), _p(wq), _p(wv), _p(kt_cache),
                                                _p(v_cache), _p(q_output), B, S, Din, wk.shape[1], _stream()),
           "mli_get_latest_kt_q_v")


def launch_qkt(q_output, kt_cache, lengths, qkt_output):
    B, D = q_output.shape
    _check(load_library().mli_qkt(_p(q_output), _p(kt_cache), _p(lengths), _p(qkt_output), B, kt_cache.shape[2], D,
                                  _stream()), "mli_qkt")


def launch_softmax_in_place_with_lengths(qkt_output, lengths):
    B, S = qkt_output.shape
    _check(load_library().mli_softmax_in_place_with_lengths(_p(qkt_output), _p(lengths), B, S, _stream()),
           "mli_softmax_in_place_with_lengths")


def launch_softmax_v(softmax_result, v_cache, attention_result, lengths):
    B, S, D = v_cache.shape
    ws, need = workspace_for(B, S, D, v_cache.device)
    _check(load_library().mli_softmax_v(_p(softmax_result), _p(v_cache), _p(lengths), _p(attention_result), B, S, D,
                                        _p(ws), need, _stream()), "mli_softmax_v")


def inference_self_attention(inp_embedding, lengths, wk, wq, wv, new_batch_idx, kt_cache, v_cache, q_output,
                             qkt_output, attention_result, n_new_items):
    B, S, Din = inp_embedding.shape
    Dout = wk.shape[1]
    ws, need = workspace_for(B, S, Dout, inp_embedding.device)
    _check(load_library().mli_inference_self_attention(
        _p(inp_embedding), _p(lengths), _p(wk), _p(wq), _p(wv), _p(new_batch_idx), _p(kt_cache), _p(v_cache),
        _p(q_output), _p(qkt_output), _p(attention_result), B, S, Din, Dout, n_new_items, _p(ws), need, _stream()),
        "mli_inference_self_attention")


def self_attention_lean(inp_embedding, lengths, wk, wq, wv, new_batch_idx, kt_cache, v_cache, q_output, attention_result,
                        n_new_items):
    """What SelfAttentionLayer runs: the contiguous composition without the qkt_output scratch (mli_self_attention_lean)."""
    B, S, Din = inp_embedding.shape
    Dout = wk.shape[1]
    ws, need = workspace_for(B, S, Dout, inp_embedding.device)
    _check(load_library().mli_self_attention_lean(
        _p(inp_embedding), _p(lengths), _p(wk), _p(wq), _p(wv), _p(new_batch_idx), _p(kt_cache), _p(v_cache),
        _p(q_output), _p(attention_result), B, S, Din, Dout, n_new_items, _p(ws), need, _stream()),
        "mli_self_attention_lean")


def decode_scan_contiguous(q_output, kt_cache, v_cache, lengths, attention_result):
    """The single-launch scan of the lean contiguous composition on its own (mli_decode_scan_contiguous)."""
    B, D, S = kt_cache.shape
    ws, need = workspace_for(B, S, D, q_output.device)
    _check(load_library().mli_decode_scan_contiguous(_p(q_output), _p(kt_cache), _p(v_cache), _p(lengths),
                                                     _p(attention_result), B, S, D, _p(ws), need, _stream()),
           "mli_decode_scan_contiguous")


# ---- paged path (page_table: int64 [B, S/16] tensor of device addresses) -----------------------
def launch_fill_new_k_v_cache_paged_attention(page_table, new_batch_idx, lengths, wk, wv, n_new_items, n_sequence):
    B = page_table.shape[0]
    _check(load_library().mli_fill_new_k_v_cache_paged(_p(page_table), _p(new_batch_idx), _p(lengths), _p(wk), _p(wv),
                                                       B, n_sequence, wk.shape[0], n_new_items, _stream()),
           "mli_fill_new_k_v_cache_paged")


# the reference's warp-tiled variant is the same MFMA kernel here
launch_fill_new_k_v_cache_paged_attention_warp_tiling = launch_fill_new_k_v_cache_paged_attention


def launch_get_latest_k_q_v_paged_attention(page_table, lengths, wk, wq, wv, q_output, n_sequence):
    B = page_table.shape[0]
    _check(load_library().mli_get_latest_k_q_v_paged(_p(page_table), _p(lengths), _p(wk), _p(wq), _p(wv),
                                                     _p(q_output), B, n_sequence, wq.shape[0], _stream()),
           "mli_get_latest_k_q_v_paged")


def launch_qkt_paged_attention(q_output, page_table, lengths, qkt_output):
    B, D = q_output.shape
    _check(load_library().mli_qkt_paged(_p(q_output), _p(page_table), _p(lengths), _p(qkt_output), B,
                                        qkt_output.shape[1], D, _stream()), "mli_qkt_paged")


def launch_softmax_v_paged_attention(softmax_result, page_table, attention_result, lengths):
    B, S = softmax_result.shape
    D = attention_result.shape[1]
    ws, need = workspace_for(B, S, D, softmax_result.device)
    _check(load_library().mli_softmax_v_paged(_p(softmax_result), _p(page_table), _p(lengths), _p(attention_result),
                                              B, S, D, _p(ws), need, _stream()), "mli_softmax_v_paged")


def paged_attention(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, qkt_output, attention_result,
                    n_new_items, n_sequence):
    B = page_table.shape[0]
    D = wk.shape[0]
    ws, need = workspace_for(B, n_sequence, D, q_output.device)
    _check(load_library().mli_paged_attention(_p(page_table), _p(lengths), _p(wk), _p(wq), _p(wv), _p(new_batch_idx),
                                              _p(q_output), _p(qkt_output), _p(attention_result), B, n_sequence, D,
                                              n_new_items, _p(ws), need, _stream()), "mli_paged_attention")


class GemmHandle:
    """Stands where the reference passes a cublasHandle_t (include/kernels/paged_attention.h:46-63); empty."""


def launch_get_latest_k_q_v_paged_attention_cublas(page_table, lengths, latest_emb, wk, wq, wv, q_output,
                                                   temp_placeholder, handle, n_sequence):
    """The reference's argument list (paged_attention.h:57-63).  latest_emb / temp_placeholder were scratch of the
    three cublasSgemm calls: accepted, never touched (one gather-GEMM-scatter kernel needs neither)."""
    assert isinstance(handle, GemmHandle) and latest_emb.shape == q_output.shape == temp_placeholder.shape
    launch_get_latest_k_q_v_paged_attention(page_table, lengths, wk, wq, wv, q_output, n_sequence)


def paged_attention_with_cublas(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, qkt_output, attention_result,
                                latest_emb=None, temp_placeholder=None, n_new_items=None, n_sequence=None, handle=None):
    """paged_attention.h:46-54 (…, latest_emb, temp_placeholder, n_new_items, n_sequence, handle); the short form
    (…, attention_result, n_new_items, n_sequence) of paged_attention is accepted too."""
    if n_sequence is None:  # called with paged_attention's positional list
        latest_emb, temp_placeholder, n_new_items, n_sequence = None, None, latest_emb, temp_placeholder
    else:
        assert isinstance(handle, GemmHandle) and latest_emb.shape == q_output.shape == temp_placeholder.shape
    paged_attention(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, qkt_output, attention_result,
                    n_new_items, n_sequence)


# ---- bf16 paged path (extension; BASELINE config 4).  page_table addresses point at bf16 pages ----------
def launch_fill_new_k_v_cache_paged_attention_bf16(page_table, new_batch_idx, lengths, wk, wv, n_new_items, n_sequence):
    B = page_table.shape[0]
    _check(load_library().mli_fill_new_k_v_cache_paged_bf16(_p(page_table), _p(new_batch_idx), _p(lengths), _p(wk),
                                                            _p(wv), B, n_sequence, wk.shape[0], n_new_items,
                                                            _stream()), "mli_fill_new_k_v_cache_paged_bf16")


def launch_get_latest_k_q_v_paged_attention_bf16(page_table, lengths, wk, wq, wv, q_output, n_sequence):
    B = page_table.shape[0]
    _check(load_library().mli_get_latest_k_q_v_paged_bf16(_p(page_table), _p(lengths), _p(wk), _p(wq), _p(wv),
                                                          _p(q_output), B, n_sequence, wq.shape[0], _stream()),
           "mli_get_latest_k_q_v_paged_bf16")


def launch_qkt_paged_attention_bf16(q_output, page_table, lengths, qkt_output):
    B, D = q_output.shape
    _check(load_library().mli_qkt_paged_bf16(_p(q_output), _p(page_table), _p(lengths), _p(qkt_output), B,
                                             qkt_output.shape[1], D, _stream()), "mli_qkt_paged_bf16")


def launch_softmax_v_paged_attention_bf16(softmax_result, page_table, attention_result, lengths):
    B, S = softmax_result.shape
    D = attention_result.shape[1]
    ws, need = workspace_for(B, S, D, softmax_result.device)
    _check(load_library().mli_softmax_v_paged_bf16(_p(softmax_result), _p(page_table), _p(lengths),
                                                   _p(attention_result), B, S, D, _p(ws), need, _stream()),
           "mli_softmax_v_paged_bf16")


def paged_attention_bf16(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, qkt_output, attention_result,
                         n_new_items, n_sequence):
    B = page_table.shape[0]
    D = wk.shape[0]
    ws, need = workspace_for(B, n_sequence, D, q_output.device)
    _check(load_library().mli_paged_attention_bf16(_p(page_table), _p(lengths), _p(wk), _p(wq), _p(wv),
                                                   _p(new_batch_idx), _p(q_output), _p(qkt_output),
                                                   _p(attention_result), B, n_sequence, D, n_new_items, _p(ws), need,
                                                   _stream()), "mli_paged_attention_bf16")


def launch_paged_attention_encoder_kernel_bf16(emb_table, wpe, inp, page_table, lengths, new_item_indices, n_new_items):
    B, S = inp.shape
    _check(load_library().mli_paged_attention_encoder_bf16(_p(emb_table), _p(wpe), _p(inp), _p(page_table),
                                                           _p(lengths), _p(new_item_indices), B, S,
                                                           emb_table.shape[1], n_new_items, _stream()),
           "mli_paged_attention_encoder_bf16")


def launch_paged_attention_decoder_multi_rounds_bf16(batch_result, emb_table, emb_score, wpe_table, page_table,
                                                     lengths, decoder_result, i_decoder):
    B, D = batch_result.shape
    n_res = decoder_result.shape[1] if decoder_result.dim() == 2 else 1
    _check(load_library().mli_paged_decoder_multi_rounds_bf16(_p(batch_result), _p(emb_table), _p(emb_score),
                                                              _p(wpe_table), _p(page_table), _p(lengths),
                                                              _p(decoder_result), B, emb_table.shape[0],
                                                              wpe_table.shape[0], D, n_res, i_decoder, _stream()),
           "mli_paged_decoder_multi_rounds_bf16")


def decode_scan_paged(q_output, page_table, lengths, qkt_output, attention_result, elem_bf16, phases=3, n_sequence=None):
    """Single-pass scores + masked softmax + softmax.V over the pages (mli_decode_scan_paged).  phases & 4 = lean mode:
    qkt_output may be None (pass n_sequence then)."""
    B, D = q_output.shape
    S = qkt_output.shape[1] if qkt_output is not None else n_sequence
    ws, need = workspace_for(B, S, D, q_output.device)
    _check(load_library().mli_decode_scan_paged(_p(q_output), _p(page_table), _p(lengths), _p(qkt_output),
                                                _p(attention_result), B, S, D, int(elem_bf16), int(phases), _p(ws),
                                                need, _stream()), "mli_decode_scan_paged")


def _elem_of(wk, elem):
    """MLI_ELEM_* of the pages: given, or from the weights' dtype (fp8 pages come with bf16 weights: say so)."""
    return int(wk.dtype == torch.bfloat16) if elem is None else int(elem)


def paged_attention_lean(page_table, lengths, wk, wq, wv, new_batch_idx, q_output, attention_result, n_new_items,
                         n_sequence, elem=None):
    """What the attention layers run: the paged composition without materialising scores / probabilities
    (mli_paged_attention_lean); page element type = elem (ELEM_*), default from the weights' dtype."""
    B = page_table.shape[0]
    D = wk.shape[0]
    ws, need = workspace_for(B, n_sequence, D, q_output.device)
    _check(load_library().mli_paged_attention_lean(_p(page_table), _p(lengths), _p(wk), _p(wq), _p(wv),
                                                   _p(new_batch_idx), _p(q_output), _p(attention_result), B, n_sequence,
                                                   D, n_new_items, _elem_of(wk, elem), _p(ws), need,
                                                   _stream()), "mli_paged_attention_lean")


def paged_prefill(emb_table, wpe, inp, page_table, lengths, new_item_indices, wk, wv, n_new_items, elem=None):
    """Encoder + prefill fill in one launch (mli_paged_prefill); page element type = elem, default from the weights."""
    B, S = inp.shape
    _check(load_library().mli_paged_prefill(_p(emb_table), _p(wpe), _p(inp), _p(page_table), _p(lengths),
                                            _p(new_item_indices), _p(wk), _p(wv), B, S, emb_table.shape[1], n_new_items,
                                            _elem_of(wk, elem), _stream()), "mli_paged_prefill")


def prefill(emb_table, wpe, inp, inp_embedding, lengths, new_item_indices, wk, wv, kt_cache, v_cache, n_new_items):
    """Encoder + prefill fill in one launch, contiguous layout (mli_prefill)."""
    B, S, Din = inp_embedding.shape
    _check(load_library().mli_prefill(_p(emb_table), _p(wpe), _p(inp), _p(inp_embedding), _p(lengths),
                                      _p(new_item_indices), _p(wk), _p(wv), _p(kt_cache), _p(v_cache), B, S, Din,
                                      wk.shape[1], n_new_items, _stream()), "mli_prefill")


_decoder_scratch = {}


def decoder_scratch_for(n_batch, n_vocab, device):
    lib = load_library()
    need = int(lib.mli_decoder_scratch_bytes(n_batch, n_vocab))
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(index).cuda_stream)
    buf = _decoder_scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.zeros(max(need, 16), dtype=torch.uint8, device=device)
        _decoder_scratch[key] = buf
    return buf, need


def decoder_fused(batch_result, emb_table, wpe_table, inp_embedding, lengths, decoder_result):
    """launch_decoder with the argmax as the logits GEMM's epilogue (no emb_score)."""
    B, D = batch_result.shape
    sc, need = decoder_scratch_for(B, emb_table.shape[0], batch_result.device)
    _check(load_library().mli_decoder_fused(_p(batch_result), _p(emb_table), _p(wpe_table), _p(inp_embedding),
                                            _p(lengths), _p(decoder_result), B, emb_table.shape[0], wpe_table.shape[0],
                                            D, _p(sc), need, _stream()), "mli_decoder_fused")


def paged_decoder_fused(batch_result, emb_table, wpe_table, page_table, lengths, decoder_result, i_decoder, elem_bf16):
    """launch_paged_attention[_cublas]_decoder_multi_rounds with the argmax as the logits GEMM's epilogue."""
    B, D = batch_result.shape
    n_res = decoder_result.shape[1] if decoder_result.dim() == 2 else 1
    sc, need = decoder_scratch_for(B, emb_table.shape[0], batch_result.device)
    _check(load_library().mli_paged_decoder_fused(_p(batch_result), _p(emb_table), _p(wpe_table), _p(page_table),
                                                  _p(lengths), _p(decoder_result), B, emb_table.shape[0],
                                                  wpe_table.shape[0], D, n_res, i_decoder, int(elem_bf16), _p(sc), need,
                                                  _stream()), "mli_paged_decoder_fused")


def stream_wait_stream(waiter, signaller):
    """torch streams: `waiter` waits for everything queued on `signaller` so far (mli_stream_wait_stream)."""
    _check(load_library().mli_stream_wait_stream(ctypes.c_void_p(waiter.cuda_stream), ctypes.c_void_p(signaller.cuda_stream)),
           "mli_stream_wait_stream")


class StepGraph:
    """hipGraph of whatever `fn` launches on the current (non-default) stream: mli_graph_* of the C ABI.  Everything
    `fn` touches must already be allocated (workspaces included): call it once eagerly first."""

    def __init__(self, fn):
        lib = load_library()
        stream = _stream()
        _check(lib.mli_graph_begin_capture(stream), "mli_graph_begin_capture")
        try:
            fn()
        finally:
            handle = ctypes.c_void_p()
            rc = lib.mli_graph_end_capture(stream, ctypes.byref(handle))
        _check(rc, "mli_graph_end_capture")
        self._exec = handle

    def launch(self):
        _check(load_library().mli_graph_launch(self._exec, _stream()), "mli_graph_launch")

    def close(self):
        if self._exec:
            load_library().mli_graph_destroy(self._exec)
            self._exec = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- encoder / decoder ---------------------------------------------------------------------
def launch_inference_optimized_encoder_kernel(emb_table, wpe, inp, inp_embedding, lengths, new_item_indices,
                                              n_new_items):
    B, S, D = inp_embedding.shape
    _check(load_library().mli_inference_optimized_encoder(_p(emb_table), _p(wpe), _p(inp), _p(inp_embedding),
                                                          _p(lengths), _p(new_item_indices), B, S, D, n_new_items,
                                                          _stream()), "mli_inference_optimized_encoder")


def launch_paged_attention_encoder_kernel(emb_table, wpe, inp, page_table, lengths, new_item_indices, n_new_items):
    B, S = inp.shape
    _check(load_library().mli_paged_attention_encoder(_p(emb_table), _p(wpe), _p(inp), _p(page_table), _p(lengths),
                                                      _p(new_item_indices), B, S, emb_table.shape[1], n_new_items,
                                                      _stream()), "mli_paged_attention_encoder")


def launch_decoder(batch_result, emb_table, emb_score, wpe_table, inp_embedding, lengths, decoder_result):
    B, D = batch_result.shape
    _check(load_library().mli_decoder(_p(batch_result), _p(emb_table), _p(emb_score), _p(wpe_table),
                                      _p(inp_embedding), _p(lengths), _p(decoder_result), B, emb_table.shape[0],
                                      wpe_table.shape[0], D, _stream()), "mli_decoder")


def launch_paged_attention_decoder_multi_rounds(batch_result, emb_table, emb_score, wpe_table, page_table, lengths,
                                                decoder_result, i_decoder):
    B, D = batch_result.shape
    n_res = decoder_result.shape[1] if decoder_result.dim() == 2 else 1
    _check(load_library().mli_paged_decoder_multi_rounds(_p(batch_result), _p(emb_table), _p(emb_score),
                                                         _p(wpe_table), _p(page_table), _p(lengths),
                                                         _p(decoder_result), B, emb_table.shape[0],
                                                         wpe_table.shape[0], D, n_res, i_decoder, _stream()),
           "mli_paged_decoder_multi_rounds")


def launch_paged_attention_cublas_decoder_multi_rounds(batch_result, emb_table, emb_score, wpe_table, page_table, lengths,
                                                       decoder_result, i_decoder, handle=None):
    """decoder.h:32-37: the same head with a cublasHandle_t at the end (a GemmHandle here; nothing to hand to MFMA)."""
    assert handle is None or isinstance(handle, GemmHandle)
    launch_paged_attention_decoder_multi_rounds(batch_result, emb_table, emb_score, wpe_table, page_table, lengths,
                                                decoder_result, i_decoder)


# ---- test / measurement support -------------------------------------------------------------
def launch_clone_inp_embedding_k_v_cache(page_table, inp_embedding, kt_cache, v_cache, lengths):
    B, S, D = inp_embedding.shape
    _check(load_library().mli_clone_inp_embedding_k_v_cache(_p(page_table), _p(inp_embedding), _p(kt_cache),
                                                            _p(v_cache), _p(lengths), B, S, D, _stream()),
           "mli_clone_inp_embedding_k_v_cache")


def stream_copy(src, dst):
    _check(load_library().mli_stream_copy(_p(src), _p(dst), src.numel(), _stream()), "mli_stream_copy")


def stream_read(src, sink):
    _check(load_library().mli_stream_read(_p(src), _p(sink), src.numel(), _stream()), "mli_stream_read")


def f32_to_fp8(src, dst=None):
    """float32 tensor -> uint8 tensor of OCP e4m3 codes with the kernels' own conversion (mli_f32_to_fp8)."""
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
    _check(load_library().mli_f32_to_fp8(_p(src), _p(dst), src.numel(), _stream()), "mli_f32_to_fp8")
    return dst


def get_latest_k_q_v_paged_lean(page_table, lengths, wk, wq, wv, q_output, n_sequence, elem=None):
    """The decode projection for any page element type (mli_get_latest_k_q_v_paged_lean)."""
    B = page_table.shape[0]
    _check(load_library().mli_get_latest_k_q_v_paged_lean(_p(page_table), _p(lengths), _p(wk), _p(wq), _p(wv), _p(q_output), B,
                                                          n_sequence, wq.shape[0], _elem_of(wk, elem), _stream()),
           "mli_get_latest_k_q_v_paged_lean")
