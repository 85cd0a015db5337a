"""Skip-list geometry and construction (host side, pure functions).

A skip list is int32 ``[batch, heads, q_tiles, k_tiles + 1]``; each row is
``[L, start_0, end_0, start_1, end_1, ...]``: L valid entries, ranges in DESCENDING tile order, both
ends inclusive for the kernel's reader (reference reader/writer at
/root/reference/hopper/_internal/cpp/mainloop_fwd_sm90_tma_gmma_ws.hpp:47-192). The functions here
restate the list bookkeeping of /root/reference/hopper/lite_attention.py:61-153, 214-242 with its
defects fixed; tile sizes always come from the kernel library.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .flash_attn_interface import get_tile_sizes, skip_list_stats


def cdiv(x: int, y: int) -> int:
    return (x + y - 1) // y


def tile_geometry(seq_len_q: int, seq_len_k: int, head_dim: int, element_size: int) -> Tuple[int, int, int, int]:
    """(block_m, block_n, q_tiles, k_tiles). k_tiles comes from the KEY length (Appendix B-2)."""
    bm, bn = get_tile_sizes(head_dim, element_size)
    return bm, bn, cdiv(seq_len_q, bm), cdiv(seq_len_k, bn)


def must_skip_row(must_skip_list: Sequence[int], block_n: int, k_tiles: int) -> List[int]:
    """README format ``[start0, end0, start1, end1, ...]`` (token indices, descending, start > end;
    /root/reference/README.md:193-197) -> list row ``[L, do-ranges...]``.

    Tile conversion follows lite_attention.py:129-138: a skip range's start rounds UP and its end
    rounds DOWN to tile indices that stay listed, so only tiles strictly inside the range are dropped.
    Unlike the reference this takes no hidden length prefix, leaves the caller's list alone and keeps
    tile k_tiles-1 (it carries the seqlen mask and is the walk's first tile)."""
    if len(must_skip_list) % 2 != 0:
        raise ValueError("must_skip_list must hold (start, end) pairs")
    edges = [k_tiles - 1]
    for i, tok in enumerate(must_skip_list):
        t = cdiv(tok, block_n) if i % 2 == 0 else tok // block_n
        edges.append(max(0, min(t, k_tiles - 1)))
    edges.append(0)
    ranges: List[int] = []
    for s, e in zip(edges[0::2], edges[1::2]):
        if s >= e:
            ranges += [s, e]
    if not ranges or ranges[0] != k_tiles - 1:
        ranges = [k_tiles - 1, k_tiles - 1] + ranges
    return [len(ranges)] + ranges


def new_skip_lists(batch: int, heads: int, q_tiles: int, k_tiles: int, device, row: Optional[Sequence[int]] = None
                   ) -> torch.Tensor:
    """Both ping-pong buffers ``[2, batch, heads, q_tiles, k_tiles + 1]``; default row ``[2, k_tiles-1, 0]``
    = one range over every tile (lite_attention.py:124, 148-151)."""
    lists = torch.zeros(2, batch, heads, q_tiles, k_tiles + 1, dtype=torch.int32, device=device)
    if row is None:
        lists[..., 0] = 2
        lists[..., 1] = k_tiles - 1
    else:
        lists[..., : len(row)] = torch.tensor(list(row), dtype=torch.int32, device=device)
    return lists


def must_do_row(must_do_list: Sequence[int], block_n: int, width: int, device) -> torch.Tensor:
    """Token ranges ``[start0, end0, ...]`` -> ONE int32 row ``[len, tiles..., 0...]`` of length ``width``.
    Starts round up, ends round down (lite_attention.py:228-235). The kernel's writer treats must-do
    ranges as (start inclusive, end exclusive) (mainloop...:154-162); ``[0, 0]`` matches nothing."""
    row = [len(must_do_list)]
    for i, tok in enumerate(must_do_list):
        row.append(cdiv(tok, block_n) if i % 2 == 0 else tok // block_n)
    if len(row) > max(width, 3):          # (a single key tile: the list row is 2 ints wide, the must-do row still [len, start, end])
        raise ValueError("must_do_list has more entries than k tiles")
    row += [0] * (max(width, 3) - len(row))
    return torch.tensor(row, dtype=torch.int32, device=device)


def listed_fraction(lists: torch.Tensor) -> float:
    """Fraction of (q-tile, k-tile) pairs a list keeps: sum over rows and ranges of (start-end+1) over
    rows*k_tiles. The statistic lite_attention.py:61-85 meant to compute (Appendix B-3). Device lists are
    reduced by the ``la_skip_list_stats`` kernel; host lists with a few tensor ops."""
    rows = lists.shape[0] * lists.shape[1] * lists.shape[2]
    k_tiles = lists.shape[3] - 1
    if rows * k_tiles <= 0:
        return 1.0
    if lists.is_cuda:
        return skip_list_stats(lists.contiguous())[0].item() / (rows * k_tiles)
    body = lists[..., 1:].to(torch.int64)
    if k_tiles % 2:
        body = torch.nn.functional.pad(body, (0, 1))
    pairs = body.unflatten(-1, (-1, 2))
    sizes = (pairs[..., 0] - pairs[..., 1] + 1).clamp_min(0)
    n_ranges = lists[..., 0].to(torch.int64).clamp_min(2) // 2      # the reader always walks range 0
    live = torch.arange(pairs.shape[-2]) < n_ranges.unsqueeze(-1)
    return (sizes * live).sum().item() / (rows * k_tiles)
