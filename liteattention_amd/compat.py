"""Adapters that put the other operator surfaces named in BASELINE.json's north_star on the ONE gfx950 kernel
(SURVEY.md §8 f2/f3). They add no new device code: every call ends in ``flash_attn_func`` above the C-ABI.

* ``fa2_*`` / ``fa3_*`` / ``flash_attn_varlen_func`` — the dense FlashAttention surfaces Wan2.x's stock
  ``flash_attention()`` wrapper calls for cross-attention (/root/reference/flash_attn/flash_attn_interface.py:998-1462;
  /root/reference/hopper/_internal/flash_attn_interface.py:487-682), with the reference's signatures and return
  conventions. Variable-length batches are ONE launch over the packed tensors (C-ABI ``cu_seqlens_q/k``), no host sync.
  Non-causal, no dropout / window / softcap / alibi: anything else raises NotImplementedError (outside the hot path).
  ``compat_shims/`` (opt-in, on PYTHONPATH) exposes them under the import names ``flash_attn`` / ``flash_attn_interface``.
* ``blockmask_to_skip_lists`` / ``flash_blocksparse_attn_func`` — a STATIC 0/1 block mask expressed as skip lists
  and run with thr=-inf (nothing new is dropped). Counterpart of the reference's FA1-era block-sparse API
  (/root/reference/flash_attn/flash_blocksparse_attn_interface.py:7-39,185-200), which is dead code there (its
  ``flash_attn_cuda.fwd_block`` exists nowhere in csrc): semantics are therefore defined here and pinned by the
  CPU oracle — block (i, j) covers queries [i*kBlockM, ...) x keys [j*kBlockN, ...) with this kernel's tiles.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .flash_attn_interface import flash_attn_func, get_tile_sizes, mha_fwd


def _to_list(x) -> List[int]:
    return x.tolist() if isinstance(x, torch.Tensor) else list(x)


def _reject(**opts):
    for name, (val, default) in opts.items():
        if isinstance(val, torch.Tensor):
            raise NotImplementedError(f"{name} (tensor argument) is outside the QK-Skip hot path of this build")
        if val != default and val is not None:
            raise NotImplementedError(f"{name}={val!r} is outside the QK-Skip hot path of this build")


def _reject_fa2(dropout_p, causal, window_size, softcap, alibi_slopes, block_table=None):
    _reject(dropout_p=(dropout_p, 0.0), causal=(causal, False), window_size=(tuple(window_size), (-1, -1)),
            softcap=(softcap, 0.0), alibi_slopes=(alibi_slopes, None), block_table=(block_table, None))


def fa2_flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                        alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """FlashAttention-2 signature (flash_attn/flash_attn_interface.py:1135-1211), dense forward on the gfx950 kernel.
    Returns ``out`` or ``(out, softmax_lse, None)``."""
    _reject_fa2(dropout_p, causal, window_size, softcap, alibi_slopes)
    if return_attn_probs:
        out, lse = flash_attn_func(q, k, v, softmax_scale=softmax_scale, return_softmax_lse=True)
        return out, lse, None
    return flash_attn_func(q, k, v, softmax_scale=softmax_scale)


def fa2_flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                 alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """kv: (batch, seqlen_k, 2, nheads_k, headdim) (flash_attn_interface.py:1057-1133); views, no copies."""
    return fa2_flash_attn_func(q, kv[:, :, 0], kv[:, :, 1], dropout_p, softmax_scale, causal, window_size, softcap,
                               alibi_slopes, deterministic, return_attn_probs)


def fa2_flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                  alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """qkv: (batch, seqlen, 3, nheads, headdim) (flash_attn_interface.py:998-1055); views, no copies."""
    return fa2_flash_attn_func(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], dropout_p, softmax_scale, causal, window_size,
                               softcap, alibi_slopes, deterministic, return_attn_probs)


def _cu_tensor(cu, device) -> torch.Tensor:
    if isinstance(cu, torch.Tensor):
        if cu.dtype != torch.int32:
            raise RuntimeError("cu_seqlens must have dtype int32")                         # flash_api.cpp cu_seqlens checks
        return cu.to(device).contiguous()
    return torch.tensor(list(cu), dtype=torch.int32, device=device)


def _varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, want_lse,
                    q_descale=None, k_descale=None, v_descale=None):
    """Packed variable-length dense attention in ONE launch: q (total_q, H, D), k/v (total_k, Hk, D), cu_seqlens_* int32
    [B+1] (device tensors are used as they are: no host sync; ``max_seqlen_*`` size the grid, as in the reference,
    hopper/_internal/flash_attn_interface.py:638-682). Returns (out bf16 (total_q, H, D), lse fp32 (H, total_q) or None)."""
    if q.dim() != 3 or k.dim() != 3 or v.dim() != 3:
        raise RuntimeError("varlen: q, k, v must be (total_tokens, nheads, headdim)")
    n_q = cu_seqlens_q.numel() if isinstance(cu_seqlens_q, torch.Tensor) else len(cu_seqlens_q)
    n_k = cu_seqlens_k.numel() if isinstance(cu_seqlens_k, torch.Tensor) else len(cu_seqlens_k)
    if n_q != n_k or n_q < 2:
        raise RuntimeError("cu_seqlens_q and cu_seqlens_k must both be [0, ..., total] with batch+1 entries")
    for cu in (cu_seqlens_q, cu_seqlens_k):
        if not isinstance(cu, torch.Tensor) and (cu[0] != 0 or any(b < a for a, b in zip(cu, cu[1:]))):
            raise RuntimeError("cu_seqlens_q and cu_seqlens_k must both be [0, ..., total] with batch+1 entries")
    if max_seqlen_q is None or max_seqlen_k is None:
        if isinstance(cu_seqlens_q, torch.Tensor) or isinstance(cu_seqlens_k, torch.Tensor):
            raise RuntimeError("max_seqlen_q / max_seqlen_k are required with device cu_seqlens (they size the launch)")
        max_seqlen_q = max(b - a for a, b in zip(cu_seqlens_q, cu_seqlens_q[1:]))
        max_seqlen_k = max(b - a for a, b in zip(cu_seqlens_k, cu_seqlens_k[1:]))
    from .flash_attn_interface import mha_fwd
    cq, ck = _cu_tensor(cu_seqlens_q, q.device), _cu_tensor(cu_seqlens_k, q.device)
    out, lse, *_ = mha_fwd(q, k, v, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=int(max_seqlen_q),
                           max_seqlen_k=int(max_seqlen_k), softmax_scale=softmax_scale, q_descale=q_descale,
                           k_descale=k_descale, v_descale=v_descale)
    return out, (lse if want_lse else None)


def fa2_flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q=None, max_seqlen_k=None,
                               dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                               alibi_slopes=None, deterministic=False, return_attn_probs=False, block_table=None):
    """FlashAttention-2 varlen signature (flash_attn/flash_attn_interface.py:1370-1462): returns ``out`` (total_q, H, D) bf16 or
    ``(out, softmax_lse (H, total_q), None)``."""
    _reject_fa2(dropout_p, causal, window_size, softcap, alibi_slopes, block_table)
    out, lse = _varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale,
                               return_attn_probs)
    return (out, lse, None) if return_attn_probs else out


def fa2_flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                                        softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                        alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """kv: (total_k, 2, nheads_k, headdim) (flash_attn_interface.py:1278-1368)."""
    return fa2_flash_attn_varlen_func(q, kv[:, 0], kv[:, 1], cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                      dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                                      return_attn_probs)


def fa2_flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                         window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                                         return_attn_probs=False):
    """qkv: (total, 3, nheads, headdim) (flash_attn_interface.py:1212-1276)."""
    return fa2_flash_attn_varlen_func(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens, cu_seqlens, max_seqlen, max_seqlen,
                                      dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                                      return_attn_probs)


def fa3_flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, seqused_q=None,
                               seqused_k=None, softmax_scale=None, causal=False, qv=None, q_descale=None, k_descale=None,
                               v_descale=None, window_size=(-1, -1), attention_chunk=0, softcap=0.0, num_splits=1,
                               pack_gqa=None, deterministic=False, sm_margin=0):
    """FlashAttention-3 varlen signature (hopper/_internal/flash_attn_interface.py:638-682); returns ``out`` like the reference
    (FlashAttnVarlenFunc.forward returns ``out`` only, :451)."""
    _reject(seqused_q=(seqused_q, None), seqused_k=(seqused_k, None), causal=(causal, False), qv=(qv, None),
            window_size=(tuple(window_size), (-1, -1)), attention_chunk=(attention_chunk, 0), softcap=(softcap, 0.0),
            pack_gqa=(pack_gqa, None))
    if num_splits not in (0, 1):
        raise NotImplementedError("split-KV is compiled out (hopper/setup.py:48)")
    out, _ = _varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale, False,
                             q_descale, k_descale, v_descale)
    return out


def fa3_flash_attn_qkvpacked_func(qkv, softmax_scale=None, causal=False, q_descale=None, k_descale=None, v_descale=None,
                                  window_size=(-1, -1), attention_chunk=0, softcap=0.0, deterministic=False, num_heads_q=None,
                                  sm_margin=0):
    """FA3 packed signature (hopper/_internal/flash_attn_interface.py:487-545): qkv (batch, seqlen, 3, nheads, headdim), or
    (batch, seqlen, nheads_q + 2*nheads_k, headdim) with ``num_heads_q``."""
    if qkv.dim() == 5:
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        if num_heads_q is None:
            raise RuntimeError("num_heads_q is required for (batch, seqlen, nheads_q + 2*nheads_k, headdim) input")
        hk = (qkv.shape[2] - num_heads_q) // 2
        q, k, v = qkv[:, :, :num_heads_q], qkv[:, :, num_heads_q:num_heads_q + hk], qkv[:, :, num_heads_q + hk:]
    return flash_attn_func(q, k, v, softmax_scale=softmax_scale, causal=causal, q_descale=q_descale, k_descale=k_descale,
                           v_descale=v_descale, window_size=window_size, attention_chunk=attention_chunk, softcap=softcap,
                           deterministic=deterministic, sm_margin=sm_margin)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q=None, max_seqlen_k=None,
                           dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                           alibi_slopes=None, deterministic=False, return_attn_probs=False, **fa3_kwargs):
    """Both varlen dialects in one entry point (kept for round-1 callers): FA2 keywords, FA3 keywords via ``**fa3_kwargs``
    (tensor-valued or non-default ones are rejected explicitly)."""
    descales = {n: fa3_kwargs.pop(n, None) for n in ("q_descale", "k_descale", "v_descale")}
    for name, val in fa3_kwargs.items():
        if isinstance(val, torch.Tensor) or val not in (None, False, 0, 0.0, 1, (-1, -1)):
            raise NotImplementedError(f"{name}={val!r} is outside the QK-Skip hot path of this build")
    _reject_fa2(dropout_p, causal, window_size, softcap, alibi_slopes)
    out, lse = _varlen_forward(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale,
                               return_attn_probs, **descales)
    return (out, lse, None) if return_attn_probs else out


# ------------------------------------------------------------------------------------------ static block masks
def blockmask_to_rows(blockmask: torch.Tensor) -> List[List[int]]:
    """One bool/0-1 mask row per q-tile over k-tiles -> list rows ``[L, start0, end0, ...]`` (descending ranges).
    Every q-tile must keep at least one k-tile (a fully masked row has no skip-list representation: the kernel's
    reader always walks its first range, mainloop_fwd_sm90_tma_gmma_ws.hpp:93-101)."""
    bm = blockmask.to(torch.bool).cpu()
    rows = []
    for m in range(bm.shape[0]):
        keep = bm[m].tolist()
        if not any(keep):
            raise ValueError(f"q-tile {m} keeps no k-tile: not representable as a skip list")
        row: List[int] = []
        j = len(keep) - 1
        while j >= 0:
            if keep[j]:
                start = j
                while j - 1 >= 0 and keep[j - 1]:
                    j -= 1
                row += [start, j]
            j -= 1
        rows.append([len(row)] + row)
    return rows


def _blockmask_to_lists_host(blockmask: torch.Tensor, k_tiles_valid: Optional[torch.Tensor] = None, validate: bool = True) -> torch.Tensor:
    """Host-side form of ``blockmask_to_lists`` for a mask that lives in host memory (tensor ops, no loop over rows)."""
    m = blockmask.to(torch.bool)
    kt = m.shape[-1]
    tile = torch.arange(kt - 1, -1, -1, device=m.device)                       # position j of the descending walk <-> tile kt-1-j
    m = m.flip(-1)
    if k_tiles_valid is not None:
        # PER BATCH (as on the device path): [B] lines up with the mask's FIRST axis whatever its rank - [B, q, k] or [B, H, q, k]; a 2-D
        # (shared) mask is expanded to one copy per batch entry
        kv = torch.as_tensor(k_tiles_valid).to(m.device).reshape(-1)
        if m.dim() == 2:
            m = m[None].expand(kv.numel(), -1, -1)
        m = m & (tile < kv.reshape([kv.numel()] + [1] * (m.dim() - 1)))
    if validate and not bool(m.any(-1).all()):
        raise ValueError("a q-tile keeps no k-tile: not representable as a skip list")
    prev = torch.nn.functional.pad(m[..., :-1], (1, 0))
    nxt = torch.nn.functional.pad(m[..., 1:], (0, 1))
    start, end = m & ~prev, m & ~nxt                                           # first / last position of every kept run
    ridx = start.cumsum(-1) - 1                                                # index of the run a position belongs to
    width = kt + 2                                                             # one dump column for the positions that write nothing
    lists = torch.zeros(*m.shape[:-1], width + 1, dtype=torch.int32, device=m.device)
    tile_b = tile.to(torch.int32).expand(m.shape)
    dump = torch.full_like(ridx, width)
    lists.scatter_(-1, torch.where(start, 1 + 2 * ridx, dump), tile_b)
    lists.scatter_(-1, torch.where(end, 2 + 2 * ridx, dump), tile_b)
    lists[..., 0] = 2 * start.sum(-1)
    return lists[..., : kt + 1].contiguous()


def blockmask_to_lists(blockmask: torch.Tensor, k_tiles_valid: Optional[torch.Tensor] = None, validate: bool = True,
                       q_tiles_valid: Optional[torch.Tensor] = None, batch: Optional[int] = None,
                       heads: Optional[int] = None) -> torch.Tensor:
    """0/1 block mask -> int32 read-list rows ``[L, start0, end0, ...]`` (descending ranges, both ends inclusive, zero padded).

    A mask on the GPU is converted by the library (``la_blockmask_to_lists``, one wave per row, no host round trip except the optional
    ``validate`` read of the empty-row counter): mask ``[q_tiles, k_tiles]`` (shared), ``[batch, q_tiles, k_tiles]`` or
    ``[batch, heads, q_tiles, k_tiles]`` -> lists ``[batch, heads, q_tiles, k_tiles + 1]`` when ``batch`` / ``heads`` are given (the mask is
    broadcast through zero strides, not materialised), else lists shaped like the mask's leading dims. ``k_tiles_valid`` /
    ``q_tiles_valid`` (int ``[batch]``): the sequence of batch b has only that many tiles — k-tiles beyond are dropped, q-tile rows beyond
    (never read by the kernel) get the whole corner. ``k_tiles_valid`` is PER BATCH on both paths: exactly ``batch`` entries (one packed
    sequence per batch index, which is what ``cu_seqlens`` describes); anything else raises ``ValueError`` whether the mask lives on the
    host or on the device. ``validate=True`` costs one blocking read of a 4-byte counter on the device path (callers on a hot path, such
    as ``flash_blocksparse_attn_qkvpacked_func`` with a mask converted once, pass ``validate=False`` after the first conversion). A mask
    in host memory goes through the tensor-op form (host bookkeeping, like ``init_skip_list``); it has no ``q_tiles_valid``."""
    if not blockmask.is_cuda:
        if q_tiles_valid is not None:
            raise NotImplementedError("q_tiles_valid: device masks only")
        if k_tiles_valid is not None and blockmask.dim() >= 3 and torch.as_tensor(k_tiles_valid).numel() != blockmask.shape[0]:
            raise ValueError("k_tiles_valid / q_tiles_valid must hold one entry per batch")       # the device path's rule
        out = _blockmask_to_lists_host(blockmask, k_tiles_valid, validate)
        if batch is not None and heads is not None and out.dim() < 4:
            out = (out[None, None] if out.dim() == 2 else out[:, None]).expand(batch, heads, -1, -1).contiguous()
        return out
    from . import _cabi
    if blockmask.dtype == torch.uint8:
        m = blockmask
    else:                                                       # bool is one byte of 0 / 1: reinterpreted, not copied
        m = (blockmask if blockmask.dtype == torch.bool else blockmask != 0).view(torch.uint8)
    lead = tuple(m.shape[:-2])
    qt, kt = m.shape[-2], m.shape[-1]
    if m.dim() > 4:
        raise ValueError("blockmask must be [q_tiles, k_tiles], [batch, q_tiles, k_tiles] or [batch, heads, q_tiles, k_tiles]")
    if m.stride(-1) != 1 or m.stride(-2) != kt:
        m = m.contiguous()
    B = batch if batch is not None else (lead[0] if len(lead) >= 1 else 1)
    H = heads if heads is not None else (lead[1] if len(lead) == 2 else 1)
    if len(lead) >= 1 and lead[0] not in (1, B) or len(lead) == 2 and lead[1] not in (1, H):
        raise ValueError("blockmask batch/heads do not match")
    sb = m.stride(0) if len(lead) >= 1 and lead[0] == B and B > 1 else 0
    sh = m.stride(1) if len(lead) == 2 and lead[1] == H and H > 1 else 0
    lists = torch.empty((B, H, qt, kt + 1), dtype=torch.int32, device=m.device)
    empty = torch.empty(1, dtype=torch.int32, device=m.device) if validate else None

    def _valid(t):
        if t is None:
            return None
        t = t.to(device=m.device, dtype=torch.int32).contiguous()
        if t.numel() != B:
            raise ValueError("k_tiles_valid / q_tiles_valid must hold one entry per batch")
        return t
    kv, qv = _valid(k_tiles_valid), _valid(q_tiles_valid)
    with torch.cuda.device(m.device):
        rc = _cabi.load().la_blockmask_to_lists(m.data_ptr(), sb, sh, B, H, qt, kt, qv.data_ptr() if qv is not None else None,
                                                kv.data_ptr() if kv is not None else None, lists.data_ptr(),
                                                empty.data_ptr() if empty is not None else None,
                                                torch.cuda.current_stream(m.device).cuda_stream)
    if rc != _cabi.LA_OK:
        raise RuntimeError(f"la_blockmask_to_lists: {_cabi.status_string(rc)}")
    if validate and int(empty.item()) != 0:
        raise ValueError("a q-tile keeps no k-tile: not representable as a skip list")
    if batch is None and heads is None:
        return lists.reshape(*lead, qt, kt + 1)
    return lists


def blockmask_to_skip_lists(blockmask: torch.Tensor, batch: int, heads: int, device) -> torch.Tensor:
    """blockmask [q_tiles, k_tiles] (shared by all batches/heads) or [batch, heads, q_tiles, k_tiles] ->
    int32 skip lists ``[2, batch, heads, q_tiles, k_tiles + 1]`` (both ping-pong buffers identical). Converted on ``device``
    (``blockmask_to_lists``: the library's kernel on a GPU, which broadcasts a shared mask through zero strides)."""
    if blockmask.dim() not in (2, 4):
        raise ValueError("blockmask must be [q_tiles, k_tiles] or [batch, heads, q_tiles, k_tiles]")
    if blockmask.dim() == 4 and tuple(blockmask.shape[:2]) != (batch, heads):
        raise ValueError("blockmask batch/heads do not match")
    lists = blockmask_to_lists(blockmask.to(device), batch=batch, heads=heads)
    return torch.stack([lists, lists]).contiguous()


def flash_blocksparse_attn_func(q, k, v, blockmask: torch.Tensor, softmax_scale=None, return_softmax_lse=False,
                                skip_lists: Optional[torch.Tensor] = None):
    """Static block-sparse attention: only tiles with blockmask == 1 are computed.
    q (B,S,H,D), k/v (B,Sk,H,D) bf16; blockmask over this kernel's (kBlockM, kBlockN) tiles. Pass the ``skip_lists``
    returned by ``blockmask_to_skip_lists`` to amortise the conversion over calls."""
    B, S, H, D = q.shape
    bm, bn = get_tile_sizes(D, q.element_size())
    qt, kt = -(-S // bm), -(-k.shape[1] // bn)
    if tuple(blockmask.shape[-2:]) != (qt, kt):
        raise ValueError(f"blockmask must be [..., {qt}, {kt}] for tiles ({bm}, {bn})")
    if skip_lists is None:
        skip_lists = blockmask_to_skip_lists(blockmask, B, H, q.device)
    must_do = torch.tensor([2, 0, 0], dtype=torch.int32, device=q.device)
    return flash_attn_func(q, k, v, softmax_scale=softmax_scale, attn_read_list=skip_lists[0],
                           attn_must_do_list=must_do, attn_write_list=skip_lists[1], thr=float("-inf"),
                           return_softmax_lse=return_softmax_lse)


def convert_blockmask(blockmask: torch.Tensor, causal: bool = False) -> torch.Tensor:
    """Name kept from the reference (flash_blocksparse_attn_interface.py:7-39), whose converted format feeds a CUDA entry point
    that exists nowhere in its csrc; here the 0/1 mask is already the kernel-facing form (it becomes skip lists)."""
    if causal:
        raise NotImplementedError("causal block-sparse attention is outside the QK-Skip hot path of this build")
    return blockmask.to(torch.bool)


def _blocksparse_packed_per_sequence(qkv, cu, mask, max_s, softmax_scale, return_attn_probs, bm, bn):
    H, D = qkv.shape[2], qkv.shape[3]
    out = torch.empty((qkv.shape[0], H, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((H, qkv.shape[0]), dtype=torch.float32, device=qkv.device) if return_attn_probs else None
    for b in range(len(cu) - 1):
        t0, t1 = cu[b], cu[b + 1]
        if t1 == t0:
            continue
        if t1 - t0 > max_s:
            raise RuntimeError("a sequence is longer than max_s")
        sub = mask[: -(-(t1 - t0) // bm), : -(-(t1 - t0) // bn)]
        res = flash_blocksparse_attn_func(qkv[t0:t1, 0][None], qkv[t0:t1, 1][None], qkv[t0:t1, 2][None], sub,
                                          softmax_scale=softmax_scale, return_softmax_lse=return_attn_probs)
        if return_attn_probs:
            out[t0:t1] = res[0][0]
            lse[:, t0:t1] = res[1][0]
        else:
            out[t0:t1] = res[0]
    return (out, lse, None) if return_attn_probs else out


def flash_blocksparse_attn_qkvpacked_func(qkv, cu_seqlens, blockmask, dropout_p, max_s, softmax_scale=None, causal=False,
                                          return_attn_probs=False, convert_mask=True):
    """The reference's signature (flash_blocksparse_attn_interface.py:185-200): qkv (total, 3, nheads, headdim) packed
    sequences, cu_seqlens [B+1], blockmask [ceil(max_s/kBlockM), ceil(max_s/kBlockN)] over THIS kernel's tiles, shared by
    every sequence and head (sequence b uses its top-left ceil(len_b/kBlockM) x ceil(len_b/kBlockN) corner). Returns
    ``context`` (total, nheads, headdim) or ``(context, softmax_lse (nheads, total), None)``.

    ONE launch for the whole packed batch (round 3; round 2 looped over the sequences after a ``.tolist()``): the mask is clipped
    to every sequence's own k-tiles and turned into list rows on the device (``blockmask_to_lists``), and ``la_fwd`` takes the lists
    together with ``cu_seqlens``. The only host round trip left is the check that no q-tile of a sequence lost all of its k-tiles
    (the reference-style ValueError); device ``cu_seqlens`` are used as they are."""
    _reject(dropout_p=(dropout_p, 0.0), causal=(causal, False))
    if qkv.dim() != 4 or qkv.shape[1] != 3:
        raise RuntimeError("qkv must be (total_tokens, 3, nheads, headdim)")
    H, D = qkv.shape[2], qkv.shape[3]
    bm, bn = get_tile_sizes(D, qkv.element_size())
    qt, kt = -(-max_s // bm), -(-max_s // bn)
    if tuple(blockmask.shape) != (qt, kt):
        raise ValueError(f"blockmask must be [{qt}, {kt}] for max_s={max_s} and tiles ({bm}, {bn})")
    mask = (convert_blockmask(blockmask, causal) if convert_mask else blockmask.to(torch.bool)).to(qkv.device)
    from . import _cabi
    if D > 128 or (_cabi.default_flags() & _cabi.LA_FLAG_KERNEL_128ROW):
        # lists + cu_seqlens in one launch exist for the hand-scheduled kernels at head_dim <= 128 (la_api.hip); elsewhere: one
        # launch per sequence after reading cu_seqlens on the host, as in round 2
        return _blocksparse_packed_per_sequence(qkv, _to_list(cu_seqlens), mask, max_s, softmax_scale, return_attn_probs, bm, bn)
    cu = cu_seqlens if torch.is_tensor(cu_seqlens) else torch.tensor(list(cu_seqlens), dtype=torch.int32)
    cu = cu.to(device=qkv.device, dtype=torch.int32).contiguous()
    B = cu.numel() - 1
    lens = cu[1:] - cu[:-1]
    if bool((lens > max_s).any()):
        raise RuntimeError("a sequence is longer than max_s")
    kt_b, qt_b = (lens + bn - 1) // bn, (lens + bm - 1) // bm
    # rows of q-tiles past a sequence's end are never read: the kernel gives them the full corner, so the "keeps a tile" check only
    # speaks about real rows
    lists = blockmask_to_lists(mask, k_tiles_valid=torch.clamp(kt_b, min=1), q_tiles_valid=qt_b, validate=True, batch=B, heads=H)
    write = torch.empty_like(lists)
    must_do = torch.tensor([2, 0, 0], dtype=torch.int32, device=qkv.device)
    out, lse, *_ = mha_fwd(qkv[:, 0], qkv[:, 1], qkv[:, 2], cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=int(max_s),
                           max_seqlen_k=int(max_s), softmax_scale=softmax_scale, attn_read_list=lists, attn_must_do_list=must_do,
                           attn_write_list=write, thr=float("-inf"), _must_do_is_1d=True)
    return (out, lse, None) if return_attn_probs else out
