"""Functional operator surface of the QK-Skip attention forward on MI355X.

Mirrors /root/reference/hopper/_internal/flash_attn_interface.py for the LiteAttention path:

* ``flash_attn_func``          same signature as the reference (:547-568), forward only
* ``FlashAttnFunc``            autograd.Function shell (:274-342); backward raises, as the reference's
                               default build compiles the backward out (hopper/setup.py:47)
* ``_flash_attn_forward``      same positional contract (:20-112): calls ``torch.ops.lite_attention.fwd``
* ``torch.ops.lite_attention.fwd``  registered here with the reference's schema
                               (hopper/_internal/cpp/flash_api.cpp:1723-1762) under the CUDA (= HIP on
                               ROCm) dispatch key; its body is ``mha_fwd`` below, the host half of
                               flash_api.cpp:667-1249, which validates like the reference, allocates
                               ``out``/``softmax_lse`` and calls the C-ABI ``la_fwd``.

There is no CPU implementation and no eager fallback: CPU tensors fail in the dispatcher, a
missing HIP library fails at import of the native binding.
"""
from __future__ import annotations

import contextlib
import ctypes
import threading
from typing import Optional, Tuple

import torch

from . import _cabi

__all__ = ["flash_attn_func", "FlashAttnFunc", "flash_attn_combine", "combine_partials", "mha_combine", "fwd_flags", "get_tile_sizes", "skip_list_stats"]

_FWD_SCHEMA = (
    "fwd("
    "Tensor q,"
    "Tensor k,"
    "Tensor v,"
    "Tensor(k_new!)? k_new = None,"
    "Tensor(v_new!)? v_new = None,"
    "Tensor? q_v = None,"
    "Tensor(out!)? out = None,"
    "Tensor? cu_seqlens_q = None,"
    "Tensor? cu_seqlens_k = None,"
    "Tensor? cu_seqlens_k_new = None,"
    "Tensor? seqused_q = None,"
    "Tensor? seqused_k = None,"
    "int? max_seqlen_q = None,"
    "int? max_seqlen_k = None,"
    "Tensor? page_table = None,"
    "Tensor? kv_batch_idx = None,"
    "Tensor? leftpad_k = None,"
    "Tensor? rotary_cos = None,"
    "Tensor? rotary_sin = None,"
    "Tensor? seqlens_rotary = None,"
    "Tensor? q_descale = None,"
    "Tensor? k_descale = None,"
    "Tensor? v_descale = None,"
    "float? softmax_scale = None,"
    "bool is_causal = False,"
    "int window_size_left = -1,"
    "int window_size_right = -1,"
    "int attention_chunk = 0,"
    "float softcap = 0.0,"
    "bool is_rotary_interleaved = False,"
    "Tensor? scheduler_metadata = None,"
    "int num_splits = 0,"
    "bool? pack_gqa = None,"
    "int sm_margin = 0,"
    "Tensor? attn_read_list = None,"
    "Tensor? attn_must_do_list = None,"
    "Tensor? attn_write_list = None,"
    "float thr = -3.0) -> (Tensor(out!), Tensor, Tensor, Tensor)"
)


_KERNEL_HEAD_DIM = {}      # (head_dim, element_size, kernel-selection flags) -> instantiated head_dim (the library's table never changes)


def kernel_head_dim(head_dim: int, element_size: int, flags: Optional[int] = None) -> int:
    """The instantiated head_dim that serves ``head_dim``: the next size up for which the library has a kernel — the host
    zero-pads q, k, v to it (``mha_fwd``), which is exact: zero columns add 0 to every score and give zero output columns,
    which are sliced away. The reference instantiates 64/96/128/192/256 (hopper/setup.py:57-61) and picks the next size up the
    same way (flash_api.cpp round_up_headdim). bf16 / fp16: 64, 96, 128, 192, 256 are built (with LA_FWD_KERNEL=v2, the
    hipcc-scheduled A/B kernels: 64, 128, 256); fp8: the same five (round 6: native bodies at all of them); beyond that the library's typed error is raised. The library is
    asked (``la_get_tile_sizes_ex``), there is no second table here."""
    if head_dim <= 0 or head_dim % (16 if element_size == 1 else 8) != 0:
        return head_dim                                      # la_get_tile_sizes / mha_fwd report the error
    flags = (_cabi.default_flags() if flags is None else flags) & _cabi.GEOMETRY_FLAGS
    key = (head_dim, element_size, flags)
    if key not in _KERNEL_HEAD_DIM:
        _KERNEL_HEAD_DIM[key] = next((d for d in (64, 96, 128, 192, 256)
                                      if d >= head_dim and _cabi.is_instantiated(d, element_size, flags)), head_dim)
    return _KERNEL_HEAD_DIM[key]


def get_tile_sizes(head_dim: int, element_size: int) -> Tuple[int, int]:
    """(kBlockM, kBlockN) of the gfx950 kernel that serves this head_dim — the single source for skip-list geometry.
    e4m3 above head_dim 128 runs one 32-row q-block per wave: (128, 64), the tiles of the bf16 kernels of those head dims."""
    flags = _cabi.default_flags()
    return _cabi.get_tile_sizes(kernel_head_dim(head_dim, element_size, flags), element_size, flags)


def device_slots(head_dim: int, element_size: int) -> Tuple[int, int]:
    """(compute units, resident workgroups per compute unit) of the kernel that serves this head_dim / element size - mapped exactly
    as ``get_tile_sizes`` maps them (a head dim between instantiations runs the next size up; e4m3 above head_dim 128 runs the bf16
    kernel of that head dim, which the library knows), THEN asked from the library (``la_device_slots`` only knows instantiated kernels: 80 -> LA_ERR_HEAD_DIM)."""
    flags = _cabi.default_flags()
    return _cabi.device_slots(kernel_head_dim(head_dim, element_size, flags), element_size, flags)


def q_tiles_per_item(head_dim: int, element_size: int) -> int:
    """q-tiles (rows of the skip lists) one workgroup item of the kernel covers: 2 under LA_FLAG_HALF_VOTE at bf16 / fp16 head dims <= 128 (lists
    per 128-row half of a 256-row workgroup: q-tile windows then start on an even q-tile and hold an even number unless they reach the
    last one), else 1."""
    flags = _cabi.default_flags()
    half = (flags & _cabi.LA_FLAG_HALF_VOTE) and not (flags & _cabi.LA_FLAG_KERNEL_128ROW) and element_size == 2 \
        and kernel_head_dim(head_dim, element_size, flags) in (64, 96, 128)
    return 2 if half else 1


def _check_list(t: Optional[torch.Tensor], name: str, q: torch.Tensor) -> Optional[int]:
    """flash_api.cpp:919-963: int32, 4-D, contiguous (same messages)."""
    if t is None:
        return None
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be int32 tensor")
    if t.dim() != 4:
        raise RuntimeError(f"{name} must be 4D tensor with shape [batch, heads, q_blocks, k_blocks]")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.device != q.device:
        raise RuntimeError(f"{name} must be on the same device as q")
    return t.data_ptr()


_tls = threading.local()


def _scoped_flags() -> int:
    return getattr(_tls, "flags", 0)


def _scoped_clear() -> int:
    return getattr(_tls, "clear", 0)


@contextlib.contextmanager
def fwd_flags(flags: int, clear: int = 0):
    """Every forward call made by this thread inside the block carries these extra ``la_fwd_args.flags`` (LA_FLAG_EXACT_RESCALE,
    LA_FLAG_FP8_MFMA_ROWSUM, LA_FLAG_FP8_ENCODED_P) - whichever surface it goes through (``LiteAttention.__call__``, ``flash_attn_func``, the
    registered op, the varlen adapters): the reference's signatures have no argument for them. ``clear``: flags taken OUT of whatever the
    host default (``LA_FP8_P``) or an outer block set: ``SeqParallelLiteAttention`` clears the two fp8 flags to get the reference's
    fp32-exact LSE for e4m3 inputs whose partial results are merged by LSE, whatever form of P the process runs with otherwise."""
    ok = _cabi.LA_FLAG_EXACT_RESCALE | _cabi.LA_FLAG_FP8_MFMA_ROWSUM | _cabi.LA_FLAG_FP8_ENCODED_P
    if (flags | clear) & ~ok:
        raise ValueError("fwd_flags accepts LA_FLAG_EXACT_RESCALE, LA_FLAG_FP8_MFMA_ROWSUM and LA_FLAG_FP8_ENCODED_P only")
    prev, prev_clear = _scoped_flags(), _scoped_clear()
    _tls.flags = (prev | flags) & ~clear
    _tls.clear = (prev_clear | clear) & ~flags
    try:
        yield
    finally:
        _tls.flags, _tls.clear = prev, prev_clear


def mha_fwd(q, k, v, k_new=None, v_new=None, q_v=None, out=None, cu_seqlens_q=None, cu_seqlens_k=None,
            cu_seqlens_k_new=None, seqused_q=None, seqused_k=None, max_seqlen_q=None, max_seqlen_k=None,
            page_table=None, kv_batch_idx=None, leftpad_k=None, rotary_cos=None, rotary_sin=None,
            seqlens_rotary=None, q_descale=None, k_descale=None, v_descale=None, softmax_scale=None,
            is_causal=False, window_size_left=-1, window_size_right=-1, attention_chunk=0, softcap=0.0,
            is_rotary_interleaved=False, scheduler_metadata=None, num_splits=0, pack_gqa=None, sm_margin=0,
            attn_read_list=None, attn_must_do_list=None, attn_write_list=None, thr=-3.0,
            _must_do_is_1d: bool = False, _q_windows=None, _window_hook=None, _static_sched: bool = False, _flags: int = 0):
    """Host half of the op (flash_api.cpp:667-1249) for the non-causal, fixed-length subset (MHA / GQA / MQA) that
    ``LiteAttention.__call__`` reaches. Returns ``(out, softmax_lse, out_accum, softmax_lse_accum)``.

    Extensions outside the reference schema (direct callers only): ``_must_do_is_1d`` (one shared must-do row) and
    ``_q_windows`` = [(first q-tile, q-tile count), ...]: the call becomes one launch per window on the current
    stream, ``_window_hook(i, out, row_begin, row_end)`` runs after window i has been enqueued (rows
    [row_begin, row_end) of ``out`` are complete once that launch is; used to all-gather early rows while later
    windows compute). ``_static_sched`` sets LA_FLAG_STATIC_SCHED (per-item workgroups instead of persistent ones: a
    collective running beside the launch gets CUs as items retire); "after_first" sets it on every window but the first
    (no collective is in flight beside window 0). ``_flags``: extra ``LA_FLAG_*`` bits ORed into ``la_fwd_args.flags`` (tests / A/B:
    LA_FLAG_EXACT_RESCALE, LA_FLAG_FP8_MFMA_ROWSUM, LA_FLAG_FP8_ENCODED_P); kernel-selection bits that change the tile geometry are not accepted here."""
    _flags |= _scoped_flags()
    if _flags & ~(_cabi.LA_FLAG_EXACT_RESCALE | _cabi.LA_FLAG_FP8_MFMA_ROWSUM | _cabi.LA_FLAG_FP8_ENCODED_P):
        raise ValueError("_flags accepts LA_FLAG_EXACT_RESCALE, LA_FLAG_FP8_MFMA_ROWSUM and LA_FLAG_FP8_ENCODED_P only")
    if not q.is_cuda:
        raise RuntimeError("lite_attention::fwd has no CPU implementation (HIP device tensors required)")
    if q.dtype not in (torch.bfloat16, torch.float16, torch.float8_e4m3fn):
        raise RuntimeError("FlashAttention only supports fp16, bf16, and fp8_e4m3 type")          # :715
    is_fp8 = q.dtype == torch.float8_e4m3fn
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError("query and key must have the same dtype")                              # :718-719
    for name, val in (("k_new", k_new), ("v_new", v_new), ("q_v", q_v), ("cu_seqlens_k_new", cu_seqlens_k_new),
                      ("seqused_q", seqused_q), ("seqused_k", seqused_k), ("page_table", page_table),
                      ("kv_batch_idx", kv_batch_idx), ("leftpad_k", leftpad_k), ("rotary_cos", rotary_cos),
                      ("rotary_sin", rotary_sin), ("seqlens_rotary", seqlens_rotary),
                      ("scheduler_metadata", scheduler_metadata)):
        if val is not None:
            raise NotImplementedError(f"{name} is outside the QK-Skip hot path (compiled out of the reference's "
                                      "default LiteAttention build, hopper/setup.py:47-63)")
    if is_causal or window_size_left >= 0 or window_size_right >= 0 or attention_chunk != 0:
        raise NotImplementedError("causal / local / chunked attention is outside the QK-Skip hot path "
                                  "(the reference's skip walk is non-causal only, mainloop:1757-1827)")
    if softcap != 0.0:
        raise NotImplementedError("softcap is compiled out (hopper/setup.py:52)")
    if num_splits < -1 or num_splits > 128:
        raise RuntimeError("num_splits must be in [-1, 128]")
    if num_splits > 1 and (attn_read_list is not None or _q_windows is not None or cu_seqlens_q is not None or cu_seqlens_k is not None):
        raise NotImplementedError("split-KV (num_splits > 1) serves dense fixed-length launches only: a skip list walks ITS tiles of the whole key range")
    if pack_gqa:
        raise NotImplementedError("pack_gqa is compiled out (hopper/setup.py:53)")
    if cu_seqlens_q is not None or cu_seqlens_k is not None:
        return _mha_fwd_varlen(q, k, v, out, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, q_descale, k_descale,
                               v_descale, softmax_scale, attn_read_list, attn_write_list, attn_must_do_list, thr, _must_do_is_1d)
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        raise RuntimeError("q, k, v must be 4D tensors (batch, seqlen, nheads, headdim)")
    if q.stride(-1) != 1 or k.stride(-1) != 1 or v.stride(-1) != 1:
        raise RuntimeError("Input tensor must have contiguous last dimension")                   # :726-728
    B, Sq, H, D = q.shape
    Bk, Sk, Hk, Dk = k.shape
    Dv = v.shape[-1]
    if (Bk, Dk) != (B, D) or tuple(v.shape[:3]) != (B, Sk, Hk):
        raise RuntimeError("k/v shape mismatch: expected k (batch, seqlen_k, nheads_k, headdim), v (batch, seqlen_k, nheads_k, headdim_v)")
    if H % Hk != 0:
        raise RuntimeError("Number of heads in key/value must divide number of heads in query")  # :777
    if D % (16 if is_fp8 else 8) != 0:
        raise RuntimeError("head_size should be a multiple of " + ("16" if is_fp8 else "8"))     # :854-856
    descales = []
    for name, t in (("q_descale", q_descale), ("k_descale", k_descale), ("v_descale", v_descale)):
        if t is None:
            descales.append(None)
            continue
        if not is_fp8:
            raise RuntimeError(f"{name} is only supported for fp8 inputs")
        if t.dtype != torch.float32 or tuple(t.shape) != (B, Hk) or t.device != q.device:          # :1003-1022
            raise RuntimeError(f"{name} must be a float32 tensor of shape (batch_size, nheads_k) on the input device")
        descales.append(t)
    if Dv != D:
        raise NotImplementedError("head_dim_v != head_dim is outside the QK-Skip hot path in this build")
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    if (num_splits > 1 or num_splits == -1) and attn_read_list is None and _q_windows is None:
        # split-KV on the host (round 6; the reference: get_num_splits / num_splits_heuristic, flash_api.cpp:437-465, heuristics.h:25-58,
        # compiled out of its default build, hopper/setup.py:48 - where num_splits = 0 therefore means 1, as it does here): a dense launch
        # with fewer (batch, head, q-tile) items than the device has workgroup slots leaves compute units idle (text queries against the
        # video keys: 80 items on 256). num_splits > 1: that many; -1 (extension): decide here by the reference's rule. One launch
        # over the splits (batch 1: the splits are the batch of a fixed-length launch; else a packed batch), then la_combine.
        n = _num_splits(B, H, Sq, Sk, D, q.element_size(), max(num_splits, 0))
        if n > 1:
            res = _mha_fwd_split_kv(q, k, v, n, out, softmax_scale, descales)
            if res is not None:
                return res

    host_flags = _cabi.default_flags()                       # the environment is read ONCE per call
    # (e4m3: native bodies at head dims 64 / 96 / 128 / 192 / 256 since round 6 - rounds 3-5 up-converted 192 / 256 here with torch elementwise passes
    # and ran the bf16 kernels; the sizes between run zero-padded on the next one, like the 2-byte types)
    D_kernel = kernel_head_dim(D, q.element_size(), host_flags)
    if D_kernel != D:
        # head_dim between the instantiated sizes: zero-pad the last dim (one extra pass over q, k, v; exact, see
        # kernel_head_dim), run the D_kernel kernel with the ORIGINAL softmax scale, slice the output
        def pad_last(t):                                     # fp8: pad the bytes (0x00 is +0.0 in e4m3)
            if is_fp8:
                return torch.nn.functional.pad(t.view(torch.uint8), (0, D_kernel - D)).view(t.dtype)
            return torch.nn.functional.pad(t, (0, D_kernel - D))
        out_dtype = torch.bfloat16 if is_fp8 else q.dtype
        if out is not None and (out.dtype != out_dtype or tuple(out.shape) != (B, Sq, H, D) or out.stride(-1) != 1):
            raise RuntimeError("For FP16/BF16 input, output must have the same dtype as inputs (BF16 for FP8 input), shape "
                               "(batch, seqlen_q, nheads, headdim_v) and a contiguous last dimension")
        qp, kp, vp = pad_last(q), pad_last(k), pad_last(v)
        res = mha_fwd(qp, kp, vp, q_descale=q_descale, k_descale=k_descale, v_descale=v_descale,
                      softmax_scale=softmax_scale, attn_read_list=attn_read_list, attn_must_do_list=attn_must_do_list,
                      attn_write_list=attn_write_list, thr=thr, _must_do_is_1d=_must_do_is_1d, _q_windows=_q_windows,
                      _static_sched=_static_sched, _flags=_flags,
                      _window_hook=None if _window_hook is None else
                      (lambda i, o, r0, r1: _window_hook(i, o[..., :D], r0, r1)))
        if out is None:
            out = torch.empty((B, Sq, H, D), dtype=out_dtype, device=q.device)      # contiguous, as the reference returns it
        out.copy_(res[0][..., :D])
        return (out, *res[1:])

    out_dtype = torch.bfloat16 if is_fp8 else q.dtype                                            # :859
    if out is None:
        out = torch.empty((B, Sq, H, Dv), dtype=out_dtype, device=q.device)                      # :872-886
    else:
        if out.dtype != out_dtype or tuple(out.shape) != (B, Sq, H, Dv) or out.stride(-1) != 1:  # :863
            raise RuntimeError("For FP16/BF16 input, output must have the same dtype as inputs (BF16 for FP8 input), shape "
                               "(batch, seqlen_q, nheads, headdim_v) and a contiguous last dimension")
    softmax_lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)                  # :887-892
    empty = torch.empty(0, dtype=torch.float32, device=q.device)

    read_ptr = _check_list(attn_read_list, "attn_read_list", q)
    write_ptr = _check_list(attn_write_list, "attn_write_list", q)
    if _must_do_is_1d and attn_must_do_list is not None:
        if attn_must_do_list.dtype != torch.int32 or attn_must_do_list.dim() != 1 or not attn_must_do_list.is_contiguous():
            raise RuntimeError("1-D attn_must_do_list must be a contiguous int32 vector")
        must_ptr = attn_must_do_list.data_ptr()
    else:
        must_ptr = _check_list(attn_must_do_list, "attn_must_do_list", q)

    block_m, block_n = _cabi.get_tile_sizes(D, q.element_size(), host_flags)
    q_tiles, k_tiles = -(-Sq // block_m), -(-Sk // block_n)
    for name, t in (("attn_read_list", attn_read_list), ("attn_write_list", attn_write_list),
                    ("attn_must_do_list", None if _must_do_is_1d else attn_must_do_list)):
        if t is not None and (t.shape[0] < B or tuple(t.shape[1:]) != (H, q_tiles, k_tiles + 1)):
            raise RuntimeError(f"{name} must have shape [>=batch, heads, q_blocks, k_blocks + 1] = "
                               f"[>={B}, {H}, {q_tiles}, {k_tiles + 1}] for tile sizes ({block_m}, {block_n}); "
                               f"got {tuple(t.shape)}")
    if _must_do_is_1d and attn_must_do_list is not None and attn_must_do_list.numel() < 3:
        raise RuntimeError("1-D attn_must_do_list needs at least [len, start, end]")

    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.dtype = _cabi.LA_DTYPE_FP8_E4M3 if is_fp8 else (_cabi.LA_DTYPE_FP16 if q.dtype == torch.float16 else _cabi.LA_DTYPE_BF16)
    for name, t in zip(("q", "k", "v"), descales):
        if t is not None:
            setattr(a, f"{name}_descale", t.data_ptr())
            setattr(a, f"{name}_descale_batch_stride", t.stride(0))
            setattr(a, f"{name}_descale_head_stride", t.stride(1))
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), softmax_lse.data_ptr()
    a.q_batch_stride, a.q_row_stride, a.q_head_stride = q.stride(0), q.stride(1), q.stride(2)
    a.k_batch_stride, a.k_row_stride, a.k_head_stride = k.stride(0), k.stride(1), k.stride(2)
    a.v_batch_stride, a.v_row_stride, a.v_head_stride = v.stride(0), v.stride(1), v.stride(2)
    a.o_batch_stride, a.o_row_stride, a.o_head_stride = out.stride(0), out.stride(1), out.stride(2)
    a.batch, a.seqlen_q, a.seqlen_k = B, Sq, Sk
    a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = H, Hk, D, Dv
    a.softmax_scale = float(softmax_scale)
    a.read_list, a.write_list, a.must_do_list = read_ptr, write_ptr, must_ptr
    a.must_do_is_1d = 1 if _must_do_is_1d else 0
    a.thr = float(thr)
    a.block_m, a.block_n = block_m, block_n
    # caller-owned scratch (the C side allocates nothing): fp8 = the pre-transposed V tiles; bf16 with lists = the ticket
    # counter of the dynamic work distribution. Freed after the launch by the caching allocator's stream-ordered reuse.
    workspace = None
    base_flags = ((host_flags & ~(_cabi.LA_FLAG_KERNEL_128ROW if is_fp8 else 0)) | _flags) & ~_scoped_clear()
    a.flags = base_flags | (_cabi.LA_FLAG_STATIC_SCHED if _static_sched is True else 0)
    need = _cabi.load().la_fwd_workspace_bytes(ctypes.byref(a))
    if need < 0:
        raise RuntimeError(f"lite_attention::fwd: {_cabi.status_string(int(need))}")
    if need > 0:
        workspace = torch.empty(int(need), dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = workspace.data_ptr(), int(need)
    windows = [(0, 0)] if _q_windows is None else [(int(b0), int(c0)) for b0, c0 in _q_windows]
    if _q_windows is not None and any(c0 <= 0 or b0 < 0 or b0 + c0 > q_tiles for b0, c0 in windows):
        raise RuntimeError(f"q-tile windows must lie inside [0, {q_tiles}) with positive counts; got {windows}")
    lib = _cabi.load()
    with torch.cuda.device(q.device):                                                             # CUDAGuard :885
        for i, (w_begin, w_count) in enumerate(windows):
            a.q_tile_begin, a.q_tile_count = w_begin, w_count
            a.flags = base_flags | (_cabi.LA_FLAG_V_PREPARED if (is_fp8 and i > 0) else 0) | \
                      (_cabi.LA_FLAG_STATIC_SCHED if (_static_sched is True or (_static_sched == "after_first" and i > 0))
                       else 0)                                                   # V^T tiles prepared by window 0
            stream = torch.cuda.current_stream(q.device).cuda_stream                              # :1219
            rc = lib.la_fwd(ctypes.byref(a), ctypes.c_void_p(stream))
            if rc != _cabi.LA_OK:
                msg = _cabi.status_string(rc)
                if rc == _cabi.LA_ERR_UNSUPPORTED:
                    raise NotImplementedError(msg)
                if rc == _cabi.LA_ERR_LAUNCH:
                    msg += f" (hipError {lib.la_last_hip_error()})"
                raise RuntimeError(f"lite_attention::fwd: {msg}")
            if _window_hook is not None:
                _window_hook(i, out, w_begin * block_m, min(Sq, (w_begin + w_count) * block_m) if w_count else Sq)
    return out, softmax_lse, empty, empty


def num_splits_heuristic(total_mblocks: int, slots: int, num_n_blocks: int, max_splits: int = 128, keys: Optional[int] = None) -> int:
    """The reference's ``num_splits_heuristic`` (hopper/_internal/cpp/heuristics.h:25-58) for the non-causal case, with the device's
    resident-workgroup slots in the place of its SM count: 1 when the items almost fill the device (>= 0.8 slots) or the key range is at
    most 4 tiles; else the smallest split count whose last-round efficiency is within 85 % of the best one. (Its other branch - split
    a K/V head that does not fit a 50 MB L2 - belongs to Hopper's cache and is not restated.)"""
    if total_mblocks >= 0.8 * slots or num_n_blocks <= 4:
        return 1
    if keys is not None and keys <= 4 * 176:
        return 1              # the reference's "num_n_blocks <= 4" is in ITS key tiles (176 keys at bf16 head_dim 128: "we never split for hdim = 128 and
                              # seqlen_k = 512", heuristics.h:39): below that many keys a split costs more launches than it fills compute units
    max_splits = min(max_splits, slots, num_n_blocks)
    eff = []
    for s in range(1, max_splits + 1):
        waves = total_mblocks * s / slots
        eff.append(waves / -(-(total_mblocks * s) // slots))
    best = max(eff)
    n = next(s for s, e in enumerate(eff, 1) if e >= 0.85 * best)
    # this device, measured (tools/debug/split_kv_probe.py, Sq = 512 against 75 088 keys, 80 items on 256 compute units): 2.70 ms unsplit,
    # 0.97 / 0.84 / 0.82 / 0.84 ms with 3 / 4 / 6 / 8 splits - an item that streams K / V alone pays the cold-HBM latency per tile, so a
    # few more items than slots (a short second round) still pays: at least 1.25 rounds of items, where the reference's rule stops at one
    return min(max_splits, max(n, -(-(5 * slots) // (4 * total_mblocks))))


def _num_splits(B, H, Sq, Sk, D, element_size, requested):
    flags = _cabi.default_flags()
    Dk = kernel_head_dim(D, element_size, flags)
    block_m, block_n = _cabi.get_tile_sizes(Dk, element_size, flags)
    block_m *= q_tiles_per_item(D, element_size)            # the half-vote form reports its 128-row LIST tile; a workgroup item is 256 rows
    k_tiles = -(-Sk // block_n)
    if requested > 1:
        return min(requested, k_tiles)
    cus, per = _cabi.device_slots(Dk, element_size, flags)
    return num_splits_heuristic(B * H * -(-Sq // block_m), cus * per, k_tiles, keys=Sk)


def _split_kv_one_sequence(q, k, v, n, chunk, out, softmax_scale, descales=(None, None, None)):
    """batch == 1 (the diffusion-inference case): the splits ARE the batch of a fixed-length launch - q with batch stride 0 (every split
    reads the same query rows: no replication), K / V with batch stride = one chunk of rows, partial O (n, Sq, H, D) and partial LSE
    (n, H, Sq) written in exactly the layout la_combine reads; a ragged last chunk is a second launch of batch 1. Two or three library
    calls and three allocations in all (the packed-batch form above costs a dozen tensor ops, which is what a 0.3 ms kernel notices).
    e4m3: the descales (1, Hk) ride with batch stride 0 like q; partial and merged O are bf16; one workspace for the prepared V^T tiles."""
    _, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    es = q.element_size()
    flags = _cabi.default_flags()
    block_m, block_n = _cabi.get_tile_sizes(D, es, flags)
    is_fp8 = q.dtype == torch.float8_e4m3fn
    o_dtype = torch.bfloat16 if is_fp8 else q.dtype
    if out is None:
        out = torch.empty((1, Sq, H, D), dtype=o_dtype, device=q.device)
    elif out.dtype != o_dtype or tuple(out.shape) != (1, Sq, H, D) or not out.is_contiguous():
        return None                                             # (a strided `out`: the unsplit launch writes it in place)
    o_part = torch.empty((n, Sq, H, D), dtype=o_dtype, device=q.device)
    lse_part = torch.empty((n, H, Sq), dtype=torch.float32, device=q.device)
    lse = torch.empty((1, H, Sq), dtype=torch.float32, device=q.device)
    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.dtype = _cabi.LA_DTYPE_FP8_E4M3 if is_fp8 else (_cabi.LA_DTYPE_FP16 if q.dtype == torch.float16 else _cabi.LA_DTYPE_BF16)
    o_cabi_dtype = _cabi.LA_DTYPE_BF16 if is_fp8 else a.dtype
    for name, t in zip(("q", "k", "v"), descales):                # (1, Hk) fp32: every split reads the same row
        if t is not None:
            setattr(a, f"{name}_descale", t.data_ptr())
            setattr(a, f"{name}_descale_batch_stride", 0)
            setattr(a, f"{name}_descale_head_stride", t.stride(1))
    a.q, a.q_batch_stride, a.q_row_stride, a.q_head_stride = q.data_ptr(), 0, q.stride(1), q.stride(2)
    a.k_row_stride, a.k_head_stride, a.k_batch_stride = k.stride(1), k.stride(2), chunk * k.stride(1)
    a.v_row_stride, a.v_head_stride, a.v_batch_stride = v.stride(1), v.stride(2), chunk * v.stride(1)
    a.o_batch_stride, a.o_row_stride, a.o_head_stride = Sq * H * D, H * D, D
    a.seqlen_q, a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = Sq, H, Hk, D, D
    a.softmax_scale = float(softmax_scale)
    a.block_m, a.block_n = block_m, block_n
    keep = (_cabi.LA_FLAG_EXACT_RESCALE | _cabi.LA_FLAG_FP8_MFMA_ROWSUM | _cabi.LA_FLAG_FP8_ENCODED_P) if is_fp8 else (_cabi.GEOMETRY_FLAGS | _cabi.LA_FLAG_EXACT_RESCALE)
    a.flags = ((flags | _scoped_flags()) & keep & ~_scoped_clear()) | _cabi.LA_FLAG_STATIC_SCHED
    full = n if Sk == n * chunk else n - 1                      # equal chunks; then the ragged one
    lib = _cabi.load()
    workspace = None
    if is_fp8:                                                  # the prepared V^T tiles of the larger of the two launches
        a.batch, a.seqlen_k = max(full, 1), chunk
        need = lib.la_fwd_workspace_bytes(ctypes.byref(a))
        if need < 0:
            raise RuntimeError(f"lite_attention::fwd (split-KV): {_cabi.status_string(int(need))}")
        workspace = torch.empty(int(need), dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = workspace.data_ptr(), int(need)
    with torch.cuda.device(q.device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream)
        for first, count, keys in ((0, full, chunk), (full, n - full, Sk - full * chunk)):
            if count == 0:
                continue
            a.batch, a.seqlen_k = count, keys
            a.k, a.v = k.data_ptr() + first * chunk * k.stride(1) * es, v.data_ptr() + first * chunk * v.stride(1) * es
            a.o, a.lse = o_part.data_ptr() + first * Sq * H * D * 2, lse_part.data_ptr() + first * H * Sq * 4
            rc = lib.la_fwd(ctypes.byref(a), stream)
            if rc != _cabi.LA_OK:
                raise RuntimeError(f"lite_attention::fwd (split-KV): {_cabi.status_string(rc)}")
        rc = lib.la_combine(o_part.data_ptr(), 1, lse_part.data_ptr(), out.data_ptr(), o_cabi_dtype, lse.data_ptr(), n, 1, Sq, H, D, stream)
    if rc != _cabi.LA_OK:
        raise RuntimeError(f"la_combine (split-KV): {_cabi.status_string(rc)}")
    empty = torch.empty(0, dtype=torch.float32, device=q.device)
    return out, lse, empty, empty


_SPLIT_CU = {}      # (B, n, Sq, Sk, chunk, device) -> (cu_seqlens_q, cu_seqlens_k): tiny device tensors, built once per shape


def _mha_fwd_split_kv(q, k, v, n, out, softmax_scale, descales):
    """Dense attention with the key range cut into ``n`` tile-aligned chunks: sequence (b, s) of a packed batch = the queries of batch b
    against chunk s of its keys - q is replicated per split (it is the small operand whenever splitting pays), K / V are the caller's
    tensors seen as packed rows (no copy) - one launch of the packed-batch kernel, then the LSE merge (la_combine). Returns None when K / V
    cannot be seen as packed rows (batch > 1 with a padded batch stride): the caller then runs the unsplit launch."""
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    for t in (k, v):
        if B > 1 and t.stride(0) != Sk * t.stride(1):
            return None
    _, block_n = get_tile_sizes(D, q.element_size())
    chunk = -(-(-(-Sk // block_n)) // n) * block_n            # keys per split: whole tiles
    n = -(-Sk // chunk)                                         # (no empty trailing split)
    if n <= 1:
        return None
    if B == 1 and kernel_head_dim(D, q.element_size()) == D and (q.dtype == torch.float8_e4m3fn or all(t is None for t in descales)):
        return _split_kv_one_sequence(q, k, v, n, chunk, out, softmax_scale, descales)
    key = (B, n, Sq, Sk, chunk, str(q.device))
    cu = _SPLIT_CU.get(key)
    if cu is None:
        while len(_SPLIT_CU) >= 16:                              # a handful of shapes per process; never grow without bound
            _SPLIT_CU.pop(next(iter(_SPLIT_CU)))
        cu_q = torch.arange(0, B * n + 1, dtype=torch.int32) * Sq
        cu_k = torch.tensor([b * Sk + min(s * chunk, Sk) for b in range(B) for s in range(n)] + [B * Sk], dtype=torch.int32)
        cu = _SPLIT_CU[key] = (cu_q.to(q.device), cu_k.to(q.device))
    q_rep = q.unsqueeze(1).expand(B, n, Sq, H, D).reshape(B * n * Sq, H, D)
    k_p = k.as_strided((B * Sk, Hk, D), (k.stride(1), k.stride(2), 1))
    v_p = v.as_strided((B * Sk, Hk, D), (v.stride(1), v.stride(2), 1))
    ds = [None if t is None else t.repeat_interleave(n, dim=0) for t in descales]
    o_p, lse_p, *_ = _mha_fwd_varlen(q_rep, k_p, v_p, None, cu[0], cu[1], Sq, chunk, ds[0], ds[1], ds[2], softmax_scale, None, None)
    o_part = o_p.view(B, n, Sq, H, D).transpose(0, 1)          # (n, B, Sq, H, D) view; mha_combine makes it contiguous (q-sized copies)
    lse_part = lse_p.view(H, B, n, Sq).permute(2, 1, 3, 0)     # logical (n, B, Sq, H), seqlen contiguous: the layout fwd_combine takes
    res, lse = mha_combine(o_part, lse_part, out=out)
    empty = torch.empty(0, dtype=torch.float32, device=q.device)
    return res, lse.transpose(1, 2).contiguous(), empty, empty


def _mha_fwd_varlen(q, k, v, out, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, q_descale, k_descale, v_descale,
                    softmax_scale, attn_read_list, attn_write_list, attn_must_do_list=None, thr=-3.0, _must_do_is_1d=False):
    """Packed variable-length batches (flash_api.cpp:672-674, 736-760): q (total_q, H, D), k/v (total_k, Hk, D), cu_seqlens_*
    int32 [B+1] on the device, max_seqlen_* size the grid. ONE launch, no host sync (fp8: plus the V^T prepare pass, which reads
    cu_seqlens_k itself). bf16 / fp16 / e4m3 at every head dim, e4m3 with descales (B, Hk). lse is (H, total_q).
    Skip lists (extension; the reference's varlen entry point has none, hopper/_internal/flash_attn_interface.py:638-682):
    ``[>= B, H, ceil(max_seqlen_q / kBlockM), ceil(max_seqlen_k / kBlockN) + 1]``, row (b, h, m) describing q-tile m of sequence b
    over THAT sequence's k-tiles - what the static block-sparse adapter needs to run a packed batch in one launch."""
    if cu_seqlens_q is None or cu_seqlens_k is None:
        raise RuntimeError("cu_seqlens_q and cu_seqlens_k must be given together")
    if (attn_read_list is None) != (attn_write_list is None):
        raise RuntimeError("attn_read_list and attn_write_list must be given together")
    if q.dtype not in (torch.bfloat16, torch.float16, torch.float8_e4m3fn):
        raise RuntimeError("FlashAttention only supports fp16, bf16, and fp8_e4m3 type")          # :715
    is_fp8 = q.dtype == torch.float8_e4m3fn
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise RuntimeError("query and key must have the same dtype")
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
        raise RuntimeError("varlen: q, k, v must be 3D tensors (total_tokens, nheads, headdim)")
    for name, cu in (("cu_seqlens_q", cu_seqlens_q), ("cu_seqlens_k", cu_seqlens_k)):
        if cu.dtype != torch.int32 or cu.dim() != 1 or not cu.is_contiguous() or cu.device != q.device:
            raise RuntimeError(f"{name} must be a contiguous int32 vector on the input device")              # :739-741
    if cu_seqlens_q.numel() != cu_seqlens_k.numel() or cu_seqlens_q.numel() < 2:
        raise RuntimeError("cu_seqlens_q and cu_seqlens_k must both have batch + 1 entries")
    if max_seqlen_q is None or max_seqlen_k is None:
        raise RuntimeError("max_seqlen_q and max_seqlen_k must be provided if cu_seqlens are provided")           # :744-746
    if q.stride(-1) != 1 or k.stride(-1) != 1 or v.stride(-1) != 1:
        raise RuntimeError("Input tensor must have contiguous last dimension")
    Tq, H, D = q.shape
    Tk, Hk, Dk = k.shape
    if Dk != D or tuple(v.shape) != (Tk, Hk, D):
        raise RuntimeError("k/v shape mismatch: expected k, v (total_k, nheads_k, headdim)")
    if H % Hk != 0:
        raise RuntimeError("Number of heads in key/value must divide number of heads in query")
    if D % (16 if is_fp8 else 8) != 0:
        raise RuntimeError("head_size should be a multiple of " + ("16" if is_fp8 else "8"))
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    B = cu_seqlens_q.numel() - 1
    descales = []
    for name, t in (("q_descale", q_descale), ("k_descale", k_descale), ("v_descale", v_descale)):
        if t is not None:
            if not is_fp8:
                raise RuntimeError(f"{name} is only supported for fp8 inputs")
            if t.dtype != torch.float32 or tuple(t.shape) != (B, Hk) or t.device != q.device:      # :1003-1022
                raise RuntimeError(f"{name} must be a float32 tensor of shape (batch_size, nheads_k) on the input device")
        descales.append(t)
    out_dtype = torch.bfloat16 if is_fp8 else q.dtype                                              # :859-863
    D_kernel = kernel_head_dim(D, q.element_size())
    if D_kernel != D:
        if out is not None and (out.dtype != out_dtype or tuple(out.shape) != (Tq, H, D) or out.stride(-1) != 1):
            raise RuntimeError("out must have the input dtype (bf16 for fp8 inputs), shape (total_q, nheads, headdim) and a contiguous last dimension")

        def pad(t):                                              # fp8: pad the bytes (0x00 is +0.0 in e4m3), as mha_fwd does
            if is_fp8:
                return torch.nn.functional.pad(t.view(torch.uint8), (0, D_kernel - D)).view(t.dtype)
            return torch.nn.functional.pad(t, (0, D_kernel - D))
        res = _mha_fwd_varlen(pad(q), pad(k), pad(v), None, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, descales[0],
                              descales[1], descales[2], softmax_scale, attn_read_list, attn_write_list, attn_must_do_list, thr, _must_do_is_1d)
        if out is None:
            out = torch.empty((Tq, H, D), dtype=out_dtype, device=q.device)        # contiguous (total_q, H, D), as the reference's
        out.copy_(res[0][..., :D])
        return (out, *res[1:])
    if out is None:
        out = torch.empty((Tq, H, D), dtype=out_dtype, device=q.device)
    elif out.dtype != out_dtype or tuple(out.shape) != (Tq, H, D) or out.stride(-1) != 1:
        raise RuntimeError("out must have the input dtype (bf16 for fp8 inputs), shape (total_q, nheads, headdim) and a contiguous last dimension")
    softmax_lse = torch.empty((H, Tq), dtype=torch.float32, device=q.device)
    empty = torch.empty(0, dtype=torch.float32, device=q.device)
    if Tq == 0 or max_seqlen_q <= 0:
        return out, softmax_lse, empty, empty
    flags = _cabi.default_flags()
    block_m, block_n = _cabi.get_tile_sizes(D, q.element_size(), flags)
    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.dtype = _cabi.LA_DTYPE_FP8_E4M3 if is_fp8 else (_cabi.LA_DTYPE_FP16 if q.dtype == torch.float16 else _cabi.LA_DTYPE_BF16)
    for name, t in zip(("q", "k", "v"), descales):
        if t is not None:
            setattr(a, f"{name}_descale", t.data_ptr())
            setattr(a, f"{name}_descale_batch_stride", t.stride(0))
            setattr(a, f"{name}_descale_head_stride", t.stride(1))
    workspace = None
    if attn_read_list is not None:
        q_tiles, k_tiles = -(-int(max_seqlen_q) // block_m), -(-max(int(max_seqlen_k), 0) // block_n)
        for name, t in (("attn_read_list", attn_read_list), ("attn_write_list", attn_write_list),
                        ("attn_must_do_list", None if _must_do_is_1d else attn_must_do_list)):
            if t is None:
                continue
            _check_list(t, name, q)
            if t.shape[0] < B or tuple(t.shape[1:]) != (H, q_tiles, k_tiles + 1):
                raise RuntimeError(f"{name} must have shape [>=batch, heads, q_blocks, k_blocks + 1] = [>={B}, {H}, {q_tiles}, "
                                   f"{k_tiles + 1}] for max_seqlen ({max_seqlen_q}, {max_seqlen_k}) and tile sizes ({block_m}, {block_n}); "
                                   f"got {tuple(t.shape)}")
        if _must_do_is_1d and attn_must_do_list is not None and (
                attn_must_do_list.dtype != torch.int32 or attn_must_do_list.dim() != 1 or attn_must_do_list.numel() < 3):
            raise RuntimeError("1-D attn_must_do_list must be a contiguous int32 vector [len, start, end, ...]")
        a.read_list, a.write_list = attn_read_list.data_ptr(), attn_write_list.data_ptr()
        a.must_do_list = None if attn_must_do_list is None else attn_must_do_list.data_ptr()
        a.must_do_is_1d = 1 if _must_do_is_1d else 0
        a.thr = float(thr)
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), softmax_lse.data_ptr()
    a.q_row_stride, a.q_head_stride = q.stride(0), q.stride(1)
    a.k_row_stride, a.k_head_stride = k.stride(0), k.stride(1)
    a.v_row_stride, a.v_head_stride = v.stride(0), v.stride(1)
    a.o_row_stride, a.o_head_stride = out.stride(0), out.stride(1)
    a.batch, a.seqlen_q, a.seqlen_k = B, int(max_seqlen_q), max(int(max_seqlen_k), 0)
    a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = H, Hk, D, D
    a.softmax_scale = float(softmax_scale)
    a.block_m, a.block_n = block_m, block_n
    flags = (flags | _scoped_flags()) & ~_scoped_clear()
    a.flags = flags & ((_cabi.LA_FLAG_FP8_MFMA_ROWSUM | _cabi.LA_FLAG_FP8_ENCODED_P | _cabi.LA_FLAG_STATIC_SCHED) if is_fp8 else
                       (_cabi.GEOMETRY_FLAGS | _cabi.LA_FLAG_EXACT_RESCALE | _cabi.LA_FLAG_STATIC_SCHED))
    a.cu_seqlens_q, a.cu_seqlens_k, a.total_q = cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(), Tq
    if attn_read_list is not None or is_fp8:      # ticket counters of the dynamic work distribution (as in the fixed-length path); fp8: + the V^T tiles
        need = _cabi.load().la_fwd_workspace_bytes(ctypes.byref(a))
        if need < 0:
            raise RuntimeError(f"lite_attention::fwd (varlen): {_cabi.status_string(int(need))}")
        if need > 0:
            workspace = torch.empty(int(need), dtype=torch.uint8, device=q.device)
            a.workspace, a.workspace_bytes = workspace.data_ptr(), int(need)
    with torch.cuda.device(q.device):
        rc = _cabi.load().la_fwd(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream(q.device).cuda_stream))
    if rc != _cabi.LA_OK:
        msg = _cabi.status_string(rc)
        if rc == _cabi.LA_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise RuntimeError(f"lite_attention::fwd (varlen): {msg}")
    return out, softmax_lse, empty, empty


# ---- op registration: torch.ops.lite_attention.fwd (flash_api.cpp:1722-1763, 1819-1824) ----------
_op_lib = None


def _register_op():
    global _op_lib
    if _op_lib is not None:
        return
    lib = torch.library.Library("lite_attention", "DEF")
    lib.define(_FWD_SCHEMA)

    def _fwd_impl(*args, **kwargs):
        return mha_fwd(*args, **kwargs)

    def _fwd_meta(q, k, v, k_new=None, v_new=None, q_v=None, out=None, cu_seqlens_q=None, *args, **kwargs):
        """Shapes and dtypes only (fake-tensor tracing / torch.compile treat the op as opaque): out has the input type, bf16 for
        e4m3 inputs (flash_api.cpp:859); softmax_lse is fp32 (B, H, Sq), or (H, total_q) for packed batches (:887-892)."""
        o_dtype = torch.bfloat16 if q.dtype == torch.float8_e4m3fn else q.dtype
        if out is None:
            out = torch.empty((*q.shape[:-1], v.shape[-1]), dtype=o_dtype, device=q.device)
        if cu_seqlens_q is not None:
            lse = torch.empty((q.shape[1], q.shape[0]), dtype=torch.float32, device=q.device)
        else:
            lse = torch.empty((q.shape[0], q.shape[2], q.shape[1]), dtype=torch.float32, device=q.device)
        empty = torch.empty(0, dtype=torch.float32, device=q.device)
        return out, lse, empty, torch.empty(0, dtype=torch.float32, device=q.device)

    lib.impl("fwd", _fwd_impl, "CUDA")
    lib.impl("fwd", _fwd_meta, "Meta")
    _op_lib = lib


_register_op()


def maybe_contiguous(x):
    return x.contiguous() if x is not None and x.stride(-1) != 1 else x


def _flash_attn_forward(q, k, v, k_new, v_new, qv, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new,
                        seqused_q, seqused_k, max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx,
                        leftpad_k, rotary_cos, rotary_sin, seqlens_rotary, q_descale, k_descale, v_descale,
                        softmax_scale, causal, window_size=(-1, -1), attention_chunk=0, softcap=0.0,
                        rotary_interleaved=True, scheduler_metadata=None, num_splits=1, pack_gqa=None,
                        sm_margin=0, attn_read_list=None, attn_must_do_list=None, attn_write_list=None,
                        thr=-3.0):
    """Same contract as the reference's _flash_attn_forward (:20-112)."""
    q, k, v = [maybe_contiguous(x) for x in (q, k, v)]
    if attn_must_do_list is not None and attn_must_do_list.dim() == 1:
        # 1-D broadcast must-do row: extension of the C-ABI (avoids the per-call 4-D repeat,
        # lite_attention.py:239-241). Not expressible in the reference schema -> direct call.
        out, softmax_lse, *rest = mha_fwd(
            q, k, v, k_new, v_new, qv, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q, seqused_k,
            max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin,
            seqlens_rotary, q_descale, k_descale, v_descale, softmax_scale, causal, window_size[0],
            window_size[1], attention_chunk, softcap, rotary_interleaved, scheduler_metadata, num_splits,
            pack_gqa, sm_margin, attn_read_list, attn_must_do_list, attn_write_list, thr, _must_do_is_1d=True)
        return (out, softmax_lse, *rest)
    out, softmax_lse, *rest = torch.ops.lite_attention.fwd(
        q, k, v, k_new, v_new, qv, out, cu_seqlens_q, cu_seqlens_k, cu_seqlens_k_new, seqused_q, seqused_k,
        max_seqlen_q, max_seqlen_k, page_table, kv_batch_idx, leftpad_k, rotary_cos, rotary_sin,
        seqlens_rotary, q_descale, k_descale, v_descale, softmax_scale, causal, window_size[0],
        window_size[1], attention_chunk, softcap, rotary_interleaved, scheduler_metadata, num_splits,
        pack_gqa, sm_margin, attn_read_list, attn_must_do_list, attn_write_list, thr=thr)
    return (out, softmax_lse, *rest)


class FlashAttnFunc(torch.autograd.Function):
    """Forward-only counterpart of the reference's FlashAttnFunc (:274-342)."""

    @staticmethod
    def forward(ctx, q, k, v, softmax_scale, causal, qv=None, q_descale=None, k_descale=None, v_descale=None,
                window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1, pack_gqa=None,
                deterministic=False, sm_margin=0, attn_read_list=None, attn_must_do_list=None,
                attn_write_list=None, thr=-3.0, return_softmax_lse=False):
        if softmax_scale is None:
            softmax_scale = (q.shape[-1] + (qv.shape[-1] if qv is not None else 0)) ** (-0.5)   # :300-301
        out, softmax_lse, *rest = _flash_attn_forward(
            q, k, v, None, None, qv, None, None, None, None, None, None, None, None, None, None, None,
            None, None, None, q_descale, k_descale, v_descale, softmax_scale, causal=causal,
            window_size=window_size, attention_chunk=attention_chunk, softcap=softcap, num_splits=num_splits,
            pack_gqa=pack_gqa, sm_margin=sm_margin, attn_read_list=attn_read_list,
            attn_must_do_list=attn_must_do_list, attn_write_list=attn_write_list, thr=thr)
        ctx.mark_non_differentiable(softmax_lse)
        return (out, softmax_lse) if return_softmax_lse else out

    @staticmethod
    def backward(ctx, dout, *args):
        raise NotImplementedError("lite_attention backward is not built (the reference's default build "
                                  "sets FLASH_ATTENTION_DISABLE_BACKWARD, hopper/setup.py:47)")


def flash_attn_func(q, k, v, softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None,
                    v_descale=None, window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1,
                    pack_gqa=None, deterministic=False, sm_margin=0, attn_read_list=None,
                    attn_must_do_list=None, attn_write_list=None, thr=-3.0, return_softmax_lse=False):
    """Same signature and return convention as the reference's flash_attn_func (:547-635).

    q: (batch, seqlen, nheads, headdim); k, v: (batch, seqlen_k, nheads, headdim). Returns ``out``
    (batch, seqlen, nheads, headdim) or ``(out, softmax_lse)`` with softmax_lse (batch, nheads, seqlen)
    fp32. With ``attn_read_list``/``attn_write_list`` the K-tile loop walks the read list and the skip
    decisions of this call are serialised into the write list (QK-Skip)."""
    return FlashAttnFunc.apply(q, k, v, softmax_scale, causal, qv, q_descale, k_descale, v_descale, window_size,
                               attention_chunk, softcap, num_splits, pack_gqa, deterministic, sm_margin,
                               attn_read_list, attn_must_do_list, attn_write_list, thr, return_softmax_lse)


def mha_combine(out_partial: torch.Tensor, lse_partial: torch.Tensor, out: Optional[torch.Tensor] = None,
                out_dtype: Optional[torch.dtype] = None):
    """``lite_attention::fwd_combine`` with the reference's contract (mha_combine, flash_api.cpp:1620-1718; schema :1787-1791):
    out_partial (num_splits, batch, seqlen, nheads, headdim), last dim contiguous; lse_partial LOGICAL shape (num_splits, batch,
    seqlen, nheads) with the seqlen dimension contiguous (``stride(-2) == 1``: physically (num_splits, batch, nheads, seqlen), i.e. the
    transposed view of what ``flash_attn_func`` returns per split, exactly what the reference's test feeds it,
    hopper/tests/test_flash_attn.py:1211-1212). Returns ``(out (batch, seqlen, nheads, headdim), softmax_lse (batch, seqlen, nheads))``,
    the latter the transposed view of a (batch, nheads, seqlen) buffer as in flash_api.cpp:1682. The reference takes fp32 partials only
    (:1631); 16-bit partials of the output type are accepted here in addition."""
    if not out_partial.is_cuda or not lse_partial.is_cuda:
        raise RuntimeError("Input tensor must be on CUDA device")
    if out_partial.dtype not in (torch.float32, torch.bfloat16, torch.float16) or lse_partial.dtype != torch.float32:
        raise RuntimeError("Attention combine function only support fp32 data type")
    if out_partial.dim() != 5 or lse_partial.dim() != 4:
        raise RuntimeError("out_partial must be (num_splits, batch, seqlen, nheads, headdim), lse_partial (num_splits, batch, seqlen, nheads)")
    if out_partial.stride(-1) != 1:
        raise RuntimeError("Input tensor must have contiguous last dimension")
    ns, B, S, H, Dv = out_partial.shape
    if ns > 256:
        raise RuntimeError("FlashAttention combine only supports num_splits at most 256")
    if tuple(lse_partial.shape) != (ns, B, S, H):
        raise RuntimeError(f"lse_partial must have shape ({ns}, {B}, {S}, {H})")
    if lse_partial.stride(-2) != 1 and S > 1:
        raise RuntimeError("LSE tensor must be contiguous in the seqlen dimension")
    if out_dtype is None:
        out_dtype = out_partial.dtype
    if out_dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise RuntimeError("Output type must be FP32, FP16 or BF16")
    if out_partial.dtype != torch.float32 and out_partial.dtype != out_dtype:
        raise RuntimeError("16-bit partial results must have the output dtype")
    if out is not None:
        if out.dtype != out_dtype or tuple(out.shape) != (B, S, H, Dv) or not out.is_cuda or out.stride(-1) != 1:
            raise RuntimeError("out must be (batch, seqlen, nheads, headdim) of the output dtype with a contiguous last dimension")
    Dp = -(-Dv // 8) * 8                                     # the kernel moves 8 elements per access (the reference pads to 4, :1651-1659)
    op = out_partial if Dp == Dv else torch.nn.functional.pad(out_partial, (0, Dp - Dv))
    op = op.contiguous()
    lp = lse_partial.transpose(-1, -2).contiguous()          # physical (num_splits, batch, nheads, seqlen); no copy for the reference's layout
    direct = out is not None and Dp == Dv and out.is_contiguous()
    res = out if direct else torch.empty((B, S, H, Dp), dtype=out_dtype, device=out_partial.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=out_partial.device)
    with torch.cuda.device(out_partial.device):
        stream = torch.cuda.current_stream(out_partial.device).cuda_stream
        rc = _cabi.load().la_combine(op.data_ptr(), int(op.dtype != torch.float32), lp.data_ptr(), res.data_ptr(),
                                     {torch.float16: _cabi.LA_DTYPE_FP16, torch.float32: _cabi.LA_DTYPE_FP32}.get(out_dtype, _cabi.LA_DTYPE_BF16),
                                     lse.data_ptr(), ns, B, S, H, Dp, ctypes.c_void_p(stream))
    if rc != _cabi.LA_OK:
        raise RuntimeError(f"la_combine: {_cabi.status_string(rc)}")
    if not direct:
        if out is not None:
            out.copy_(res[..., :Dv])
            res = out
        elif Dp != Dv:
            res = res[..., :Dv]
    return res, lse.transpose(1, 2)


def _combine_meta(out_partial, lse_partial, out=None, out_dtype=None):
    ns, B, S, H, Dv = out_partial.shape
    if out is None:
        out = torch.empty((B, S, H, Dv), dtype=out_dtype or out_partial.dtype, device=out_partial.device)
    return out, torch.empty((B, H, S), dtype=torch.float32, device=out_partial.device).transpose(1, 2)


def _register_combine_op():
    """``lite_attention::fwd_combine`` beside ``fwd`` (flash_api.cpp:1787-1791, 1822)."""
    _op_lib.define(_COMBINE_SCHEMA)
    _op_lib.impl("fwd_combine", mha_combine, "CUDA")
    _op_lib.impl("fwd_combine", _combine_meta, "Meta")


_COMBINE_SCHEMA = "fwd_combine(Tensor out_partial, Tensor lse_partial, Tensor(out!)? out = None, ScalarType? out_dtype = None) -> (Tensor(out!), Tensor)"
_register_combine_op()


def combine_partials(outs, lses, out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, return_lse: bool = True):
    """LSE-weighted merge of SEPARATE partial results - the ``(out, lse)`` pairs that ``flash_attn_func`` / ``LiteAttention.__call__`` return
    with ``return_softmax_lse=True``, e.g. the v2t and v2v calls of the reference's text + video recipe (/root/reference/README.md:225-246,
    which leaves the merge to the caller): ``outs`` = sequence of (batch, seqlen, nheads, headdim) tensors of one dtype (bf16 / fp16 / fp32),
    ``lses`` = sequence of fp32 (batch, nheads, seqlen). C-ABI ``la_combine_list``: the partials are read where they are (no stacking copy),
    at most 8 of them. Returns ``out`` or ``(out, lse (batch, nheads, seqlen))``."""
    outs, lses = list(outs), list(lses)
    if len(outs) != len(lses) or not 1 <= len(outs) <= 8:
        raise RuntimeError("combine_partials takes 1..8 (out, lse) pairs")
    o0 = outs[0]
    if not o0.is_cuda:
        raise RuntimeError("combine_partials has no CPU implementation")
    if o0.dim() != 4 or o0.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise RuntimeError("partials must be (batch, seqlen, nheads, headdim) fp32, bf16 or fp16 tensors")
    B, S, H, Dv = o0.shape
    if Dv % 8 != 0:
        raise RuntimeError("head_size should be a multiple of 8")
    outs = [o if o.is_contiguous() else o.contiguous() for o in outs]
    lses = [x if x.is_contiguous() else x.contiguous() for x in lses]
    for o, x in zip(outs, lses):
        if tuple(o.shape) != (B, S, H, Dv) or o.dtype != o0.dtype or o.device != o0.device:
            raise RuntimeError("all partial outputs must have one shape, dtype and device")
        if tuple(x.shape) != (B, H, S) or x.dtype != torch.float32 or x.device != o0.device:
            raise RuntimeError("every lse must be fp32 (batch, nheads, seqlen) on the device of the partials")
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else o0.dtype
    if o0.dtype != torch.float32 and o0.dtype != out_dtype:
        raise RuntimeError("16-bit partial results must have the output dtype")
    if out is None:
        out = torch.empty((B, S, H, Dv), dtype=out_dtype, device=o0.device)
    elif out.dtype != out_dtype or tuple(out.shape) != (B, S, H, Dv) or not out.is_contiguous():
        raise RuntimeError("out must be a contiguous (batch, seqlen, nheads, headdim) tensor of the output dtype")
    lse = torch.empty((B, H, S), dtype=torch.float32, device=o0.device) if return_lse else None
    n = len(outs)
    o_ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    l_ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in lses])
    with torch.cuda.device(o0.device):
        stream = torch.cuda.current_stream(o0.device).cuda_stream
        rc = _cabi.load().la_combine_list(o_ptrs, int(o0.dtype != torch.float32), l_ptrs, out.data_ptr(),
                                          {torch.float16: _cabi.LA_DTYPE_FP16, torch.float32: _cabi.LA_DTYPE_FP32}.get(out_dtype, _cabi.LA_DTYPE_BF16),
                                          None if lse is None else lse.data_ptr(), n, B, S, H, Dv, ctypes.c_void_p(stream))
    if rc != _cabi.LA_OK:
        raise RuntimeError(f"la_combine_list: {_cabi.status_string(rc)}")
    return (out, lse) if return_lse else out


def flash_attn_combine(out_partial: torch.Tensor, lse_partial: torch.Tensor, out: Optional[torch.Tensor] = None,
                       out_dtype: Optional[torch.dtype] = None, return_lse: bool = True):
    """LSE-weighted merge of per-split partial results (sequence-parallel K/V splits): the reference's flash_attn_combine
    (hopper/_internal/flash_attn_interface.py:684-685), a call of the ``lite_attention::fwd_combine`` op.

    out_partial: (num_splits, batch, seqlen, nheads, headdim) fp32, bf16 or fp16. lse_partial, fp32, in either of two layouts, and
    the returned LSE has the layout of the input:
      * the reference op's: logical (num_splits, batch, seqlen, nheads) with the seqlen dimension contiguous -> lse (batch, seqlen, nheads);
      * what ``flash_attn_func(..., return_softmax_lse=True)`` returns, stacked: contiguous (num_splits, batch, nheads, seqlen) -> lse
        (batch, nheads, seqlen). (When seqlen == nheads the strides tell them apart.)
    ``out_dtype``: default = the dtype of the partials, as in the reference (fp32 partials -> fp32 result); fp32 partials may also be merged
    into bf16 / fp16."""
    if isinstance(out_partial, (list, tuple)):               # separate partial tensors (extension): merged where they are, la_combine_list
        return combine_partials(out_partial, lse_partial, out, out_dtype, return_lse)
    if not out_partial.is_cuda:
        raise RuntimeError("flash_attn_combine has no CPU implementation")
    if out_dtype is None and out is not None:
        out_dtype = out.dtype
    if out_partial.dim() != 5 or lse_partial.dim() != 4:
        raise RuntimeError("out_partial must be (num_splits, batch, seqlen, nheads, headdim) and lse_partial 4-D")
    ns, B, S, H, Dv = out_partial.shape
    ref_layout = tuple(lse_partial.shape) == (ns, B, S, H) and (lse_partial.stride(-2) == 1 or S == 1)
    own_layout = tuple(lse_partial.shape) == (ns, B, H, S) and (lse_partial.stride(-1) == 1 or S == 1)
    if own_layout and (not ref_layout or S != 1 and lse_partial.stride(-1) == 1):
        res, lse = torch.ops.lite_attention.fwd_combine(out_partial, lse_partial.transpose(-1, -2), out, out_dtype)
        return (res, lse.transpose(1, 2)) if return_lse else res
    if not ref_layout:
        raise RuntimeError("lse_partial must have shape (num_splits, batch, nheads, seqlen), or the reference op's "
                           "(num_splits, batch, seqlen, nheads) with the seqlen dimension contiguous")
    res, lse = torch.ops.lite_attention.fwd_combine(out_partial, lse_partial, out, out_dtype)
    return (res, lse) if return_lse else res


def skip_list_stats(skip_list: torch.Tensor, batch: Optional[int] = None) -> torch.Tensor:
    """Device-side count of listed tiles: returns int64[2] = (listed tiles, rows). No host sync."""
    if skip_list.dtype != torch.int32 or skip_list.dim() != 4 or not skip_list.is_contiguous():
        raise RuntimeError("skip list must be a contiguous int32 tensor [batch, heads, q_blocks, k_blocks + 1]")
    if not skip_list.is_cuda:
        raise RuntimeError("skip_list_stats has no CPU implementation")
    nb = skip_list.shape[0] if batch is None else batch
    out = torch.empty(2, dtype=torch.int64, device=skip_list.device)
    with torch.cuda.device(skip_list.device):
        stream = torch.cuda.current_stream(skip_list.device).cuda_stream
        rc = _cabi.load().la_skip_list_stats(skip_list.data_ptr(), nb, skip_list.shape[1], skip_list.shape[2],
                                             skip_list.shape[3] - 1, out.data_ptr(), ctypes.c_void_p(stream))
    if rc != _cabi.LA_OK:
        raise RuntimeError(f"la_skip_list_stats: {_cabi.status_string(rc)}")
    return out
