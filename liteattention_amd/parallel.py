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
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from .lite_attention import LiteAttention


def head_range(num_heads: int, world: int, rank: int) -> Tuple[int, int]:
    if num_heads % world != 0:
        raise ValueError(f"num_heads={num_heads} must be divisible by the number of ranks ({world})")
    per = num_heads // world
    return rank * per, (rank + 1) * per


class HeadShardedLiteAttention:
    """LiteAttention on this rank's heads + all-gather of the outputs.

    ``__call__(q, k, v)`` takes the LOCAL head shard ``(B, S, H/G, D)`` (use ``shard()`` on full tensors) and
    returns the gathered ``(G, B, S, H/G, D)`` output (or the local output when ``gather=False`` / G == 1)."""

    def __init__(self, num_heads: int, enable_skipping: bool = True, threshold: float = -10.0,
                 max_batch_size: int = 4, process_group=None,
                 attention_fn: Optional[Callable[..., torch.Tensor]] = None):
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

    def shard(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S, H, D) -> this rank's heads (a view)."""
        assert x.shape[2] == self.num_heads
        return x[:, :, self.h0:self.h1]

    def __call__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None,
                 gather: bool = True, _kernel_events=None, **kw) -> torch.Tensor:
        assert q.shape[2] == self.h1 - self.h0, "pass the local head shard (see shard())"
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
    def to_bshd(gathered: torch.Tensor) -> torch.Tensor:
        """(G, B, S, H/G, D) -> (B, S, H, D)."""
        G, B, S, Hl, D = gathered.shape
        return gathered.permute(1, 2, 0, 3, 4).reshape(B, S, G * Hl, D)

    def reset_skip_state(self):
        self.local.reset_skip_state()

    def set_threshold(self, threshold: float):
        self.local.set_threshold(threshold)
