"""Head-sharded QK-Skip attention over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference ships no multi-GPU code on this path (``SeqParallelLiteAttention`` is bookkeeping only,
/root/reference/hopper/lite_attention.py:322-345); the unit of independence is the head: every workgroup
touches one head and the lists are indexed [b, h, q-tile, .] (mainloop_fwd_sm90_tma_gmma_ws.hpp:63-69).
So heads are partitioned — rank r owns heads [r*H/G, (r+1)*H/G) of q, k, v and its OWN skip state, which
never moves — and the only exchange is one all-gather of the bf16 output shard.

Layout of the gathered output: ``(G, B, S, H/G, D)`` — rank-major, i.e. head-major blocks. That is what
``all_gather_into_tensor`` produces with no extra copy; xGMI is a point-to-point mesh, each rank pushes its
shard to its 7 peers directly. ``to_bshd`` materialises the reference layout ``(B, S, H, D)`` when a consumer
needs it (one permute copy; a fused o-proj can read the head-major blocks directly).

Overlap (``overlap_windows`` > 1). One xGMI link moves a rank's whole output shard in about 1/5 of the time the
attention itself takes at every G in {2, 4, 8} (shard and work both shrink with G; a pair of GPUs shares ONE
link), so a gather that starts after the kernel costs ~20 % of the step. The attention is therefore issued as
several launches over q-tile WINDOWS of the same problem (C-ABI ``q_tile_begin/q_tile_count``): rows of window
i are final when its launch completes, and their all-gather runs on RCCL's stream while window i+1 computes;
only the last window's gather is exposed. Windows are whole numbers of workgroup ROUNDS (256 CUs x workgroups
per CU), so splitting the launch adds no partially filled round. The result is then a list of per-window
gathered blocks ``[(G, B, rows_i, H/G, D), ...]`` (row-chunked, which is what a row-wise consumer such as the
output projection wants); ``to_bshd`` accepts it too.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple, Union

import torch

from .lite_attention import LiteAttention


def head_range(num_heads: int, world: int, rank: int) -> Tuple[int, int]:
    if num_heads % world != 0:
        raise ValueError(f"num_heads={num_heads} must be divisible by the number of ranks ({world})")
    per = num_heads // world
    return rank * per, (rank + 1) * per


def plan_q_windows(q_tiles: int, workgroups_per_q_tile: int, n_windows: int, slots: int = 256) -> List[Tuple[int, int]]:
    """Split ``q_tiles`` into about ``n_windows`` windows ``(first q-tile, count)`` whose launches are whole numbers of
    workgroup rounds: a launch of ``count`` q-tiles has ``count * workgroups_per_q_tile`` workgroups and the chip runs
    ``slots`` at a time, so every window but the last is sized ``floor(k * slots / workgroups_per_q_tile)`` q-tiles
    (k rounds, < 1 workgroup-row of slack); the last takes the remainder (the tail a single launch would have too)."""
    if n_windows <= 1 or q_tiles < 2:
        return [(0, q_tiles)]
    rounds_total = q_tiles * workgroups_per_q_tile / slots
    k = max(1, round(rounds_total / n_windows))
    per = max(1, (k * slots) // workgroups_per_q_tile)
    windows, begin = [], 0
    while q_tiles - begin > per + per // 2:           # a remainder below half a window is merged into the last one
        windows.append((begin, per))
        begin += per
    windows.append((begin, q_tiles - begin))
    return windows


class HeadShardedLiteAttention:
    """LiteAttention on this rank's heads + all-gather of the outputs.

    ``__call__(q, k, v)`` takes the LOCAL head shard ``(B, S, H/G, D)`` (use ``shard()`` on full tensors) and
    returns the gathered ``(G, B, S, H/G, D)`` output (or the local output when ``gather=False`` / G == 1)."""

    def __init__(self, num_heads: int, enable_skipping: bool = True, threshold: float = -10.0,
                 max_batch_size: int = 4, process_group=None,
                 attention_fn: Optional[Callable[..., torch.Tensor]] = None, overlap_windows: int = 1,
                 windowed_attention_fn: Optional[Callable[..., torch.Tensor]] = None,
                 q_tile_rows: Optional[int] = None):
        self.group = process_group
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        else:
            self.world, self.rank = 1, 0
        self.num_heads = num_heads
        self.h0, self.h1 = head_range(num_heads, self.world, self.rank)
        self.local = LiteAttention(enable_skipping, threshold, max_batch_size)
        # test seam: CPU/gloo tests of the sharding + collective replace the device op with a stand-in
        self._attention = attention_fn if attention_fn is not None else self.local
        # windowed form: fn(q, k, v, q_windows, window_hook, scale) -> out, calling window_hook(i, out, row0, row1) as
        # each window's rows become final in stream order (LiteAttention.call_windowed)
        self._windowed = windowed_attention_fn if windowed_attention_fn is not None else (
            None if attention_fn is not None else
            (lambda q, k, v, windows, hook, scale, **kw: self.local.call_windowed(q, k, v, windows, hook, scale, **kw)))
        self.overlap_windows = int(overlap_windows)
        self._q_tile_rows = q_tile_rows                # test seam (CPU stand-ins have no kernel tile); None = ask the library

    def shard(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S, H, D) -> this rank's heads (a view)."""
        assert x.shape[2] == self.num_heads
        return x[:, :, self.h0:self.h1]

    def q_windows(self, q: torch.Tensor) -> List[Tuple[int, int]]:
        """The q-tile windows ``__call__`` uses for this query shape when ``overlap_windows`` > 1."""
        if self._q_tile_rows is not None:
            bm = self._q_tile_rows
        else:
            from .flash_attn_interface import get_tile_sizes
            bm, _ = get_tile_sizes(q.shape[-1], q.element_size())
        slots = 256 if bm == 256 else 512              # x64: one workgroup per CU; 128-row kernels: two
        return plan_q_windows(-(-q.shape[1] // bm), q.shape[0] * q.shape[2], self.overlap_windows, slots)

    def _call_overlapped(self, q, k, v, scale, _kernel_events, **kw) -> List[torch.Tensor]:
        import torch.distributed as dist
        B, S, Hl, D = q.shape
        windows = self.q_windows(q)
        blocks, works, offset, flat = [], [], [0], [None]

        def hook(i, out, r0, r1):
            if flat[0] is None:                        # one buffer for all windows, in the op's output dtype (bf16)
                flat[0] = torch.empty(self.world * out.numel(), dtype=out.dtype, device=out.device)
            part = out[:, r0:r1]
            if not part.is_contiguous():               # B > 1: rows of a window are not one slab
                part = part.contiguous()
            n = part.numel()
            dst = flat[0][offset[0]: offset[0] + self.world * n]
            offset[0] += self.world * n
            # enqueued on the collective's own stream behind everything issued so far on the current stream — i.e.
            # behind window i, not behind window i+1, which is launched next and overlaps with this transfer
            works.append(dist.all_gather_into_tensor(dst.view(self.world * B, r1 - r0, Hl, D), part, group=self.group,
                                                     async_op=True))
            blocks.append(dst.view(self.world, B, r1 - r0, Hl, D))

        if _kernel_events is not None:
            _kernel_events[0].record()
        self._windowed(q, k, v, windows, hook, scale, **kw)
        if _kernel_events is not None:
            _kernel_events[1].record()
        for w in works:
            w.wait()                                   # the current stream waits for the gathers; no host sync
        return blocks

    def __call__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None,
                 gather: bool = True, _kernel_events=None, **kw) -> Union[torch.Tensor, List[torch.Tensor]]:
        assert q.shape[2] == self.h1 - self.h0, "pass the local head shard (see shard())"
        if self.world > 1 and gather and self.overlap_windows > 1 and self._windowed is not None:
            return self._call_overlapped(q, k, v, scale, _kernel_events, **kw)
        if _kernel_events is not None:
            _kernel_events[0].record()
        out = self._attention(q, k, v, scale, **kw)
        if _kernel_events is not None:
            _kernel_events[1].record()
        if self.world == 1 or not gather:
            return out
        import torch.distributed as dist
        out = out.contiguous()
        gathered = torch.empty((self.world, *out.shape), dtype=out.dtype, device=out.device)
        # concatenated-along-dim-0 form (same memory as the stacked view; gloo only accepts this one)
        dist.all_gather_into_tensor(gathered.view(-1, *out.shape[1:]), out, group=self.group)
        return gathered

    @staticmethod
    def to_bshd(gathered: Union[torch.Tensor, Sequence[torch.Tensor]]) -> torch.Tensor:
        """(G, B, S, H/G, D), or the per-window list of (G, B, rows_i, H/G, D) blocks -> (B, S, H, D)."""
        if not isinstance(gathered, torch.Tensor):
            return torch.cat([HeadShardedLiteAttention.to_bshd(blk) for blk in gathered], dim=1)
        G, B, S, Hl, D = gathered.shape
        return gathered.permute(1, 2, 0, 3, 4).reshape(B, S, G * Hl, D)

    def reset_skip_state(self):
        self.local.reset_skip_state()

    def set_threshold(self, threshold: float):
        self.local.set_threshold(threshold)
