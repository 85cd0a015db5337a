"""Adapters that put the other operator surfaces named in BASELINE.json's north_star on the ONE gfx950 kernel
(SURVEY.md §8 f2/f3). They add no new device code: every call ends in ``flash_attn_func`` above the C-ABI.

* ``fa2_flash_attn_func`` / ``flash_attn_varlen_func`` — the dense FlashAttention surfaces Wan2.x's stock
  ``flash_attention()`` wrapper calls for cross-attention (/root/reference/flash_attn/flash_attn_interface.py:1135,
  1370; /root/reference/hopper/_internal/flash_attn_interface.py:638). Non-causal, no dropout / window / softcap /
  alibi: anything else raises NotImplementedError (outside the hot path).
* ``blockmask_to_skip_lists`` / ``flash_blocksparse_attn_func`` — a STATIC 0/1 block mask expressed as skip lists
  and run with thr=-inf (nothing new is dropped). Counterpart of the reference's FA1-era block-sparse API
  (/root/reference/flash_attn/flash_blocksparse_attn_interface.py:7-39,185-200), which is dead code there (its
  ``flash_attn_cuda.fwd_block`` exists nowhere in csrc): semantics are therefore defined here and pinned by the
  CPU oracle — block (i, j) covers queries [i*kBlockM, ...) x keys [j*kBlockN, ...) with this kernel's tiles.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .flash_attn_interface import flash_attn_func, get_tile_sizes


def _reject(**opts):
    for name, (val, default) in opts.items():
        if val != default and val is not None:
            raise NotImplementedError(f"{name}={val!r} is outside the QK-Skip hot path of this build")


def fa2_flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                        alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """FlashAttention-2 signature (flash_attn/flash_attn_interface.py:1135-1211), dense forward on the gfx950 kernel."""
    _reject(dropout_p=(dropout_p, 0.0), causal=(causal, False), window_size=(tuple(window_size), (-1, -1)),
            softcap=(softcap, 0.0), alibi_slopes=(alibi_slopes, None))
    if return_attn_probs:
        out, lse = flash_attn_func(q, k, v, softmax_scale=softmax_scale, return_softmax_lse=True)
        return out, lse, None
    return flash_attn_func(q, k, v, softmax_scale=softmax_scale)


def _to_list(x) -> List[int]:
    return x.tolist() if isinstance(x, torch.Tensor) else list(x)


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q=None, max_seqlen_k=None,
                           dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                           alibi_slopes=None, deterministic=False, return_attn_probs=False, **fa3_kwargs):
    """Packed variable-length attention: q (total_q, H, D), k/v (total_k, H, D), cu_seqlens_* int32 [B+1].

    Each sequence is one dense launch on a view of the packed tensors (no copies). ``cu_seqlens`` given as
    device tensors are read back once (one host sync per call) — pass Python lists to avoid it."""
    _reject(dropout_p=(dropout_p, 0.0), causal=(causal, False), window_size=(tuple(window_size), (-1, -1)),
            softcap=(softcap, 0.0), alibi_slopes=(alibi_slopes, None))
    for name, val in fa3_kwargs.items():
        if val not in (None, False, 0, 0.0, 1, (-1, -1)):
            raise NotImplementedError(f"{name}={val!r} is outside the QK-Skip hot path of this build")
    cq, ck = _to_list(cu_seqlens_q), _to_list(cu_seqlens_k)
    if len(cq) != len(ck) or cq[0] != 0 or ck[0] != 0:
        raise RuntimeError("cu_seqlens_q and cu_seqlens_k must both be [0, ..., total] with batch+1 entries")
    out = torch.empty_like(q)
    lse = torch.full((q.shape[1], q.shape[0]), float("inf"), dtype=torch.float32, device=q.device) if return_attn_probs else None
    for b in range(len(cq) - 1):
        q0, q1, k0, k1 = cq[b], cq[b + 1], ck[b], ck[b + 1]
        if q1 == q0:
            continue
        res = flash_attn_func(q[q0:q1].unsqueeze(0), k[k0:k1].unsqueeze(0), v[k0:k1].unsqueeze(0),
                              softmax_scale=softmax_scale, return_softmax_lse=return_attn_probs)
        if return_attn_probs:
            out[q0:q1] = res[0][0]
            lse[:, q0:q1] = res[1][0]
        else:
            out[q0:q1] = res[0]
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


def blockmask_to_skip_lists(blockmask: torch.Tensor, batch: int, heads: int, device) -> torch.Tensor:
    """blockmask [q_tiles, k_tiles] (shared by all batches/heads) or [batch, heads, q_tiles, k_tiles] ->
    int32 skip lists ``[2, batch, heads, q_tiles, k_tiles + 1]`` (both ping-pong buffers identical)."""
    if blockmask.dim() == 2:
        blockmask = blockmask[None, None].expand(batch, heads, -1, -1)
    B, H, Qt, Kt = blockmask.shape
    if (B, H) != (batch, heads):
        raise ValueError("blockmask batch/heads do not match")
    lists = torch.zeros(B, H, Qt, Kt + 1, dtype=torch.int32)
    cache = {}
    for b in range(B):
        for h in range(H):
            key = blockmask[b, h].to(torch.bool).cpu().numpy().tobytes()
            if key not in cache:
                rows = blockmask_to_rows(blockmask[b, h])
                t = torch.zeros(Qt, Kt + 1, dtype=torch.int32)
                for m, r in enumerate(rows):
                    t[m, : len(r)] = torch.tensor(r, dtype=torch.int32)
                cache[key] = t
            lists[b, h] = cache[key]
    return torch.stack([lists, lists]).to(device).contiguous()


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
