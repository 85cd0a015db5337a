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
only the last window's gather is exposed. Windows are whole numbers of workgroup ROUNDS (compute units x workgroups
per CU, asked from the library: ``la_device_slots``), so splitting the launch adds no partially filled round. The result is then a list of per-window
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
                 q_tile_rows: Optional[int] = None, _collective_at_world_1: bool = False, slots: Optional[int] = None):
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
            # static_sched: per-item workgroups release a CU every item, so RCCL's kernels get in beside the next window;
            # persistent workgroups would hold every CU until their window ends and push each gather one window late
            (lambda q, k, v, windows, hook, scale, **kw: self.local.call_windowed(q, k, v, windows, hook, scale,
                                                                                 static_sched="after_first", **kw)))
        self.overlap_windows = int(overlap_windows)
        self._q_tile_rows = q_tile_rows                # test seam (CPU stand-ins have no kernel tile); None = ask the library
        self._slots = slots                            # with q_tile_rows: resident workgroups to plan windows for (default 256)
        # test seam: run the collective path on a 1-rank group too (an RCCL all-gather of one rank is a copy on RCCL's
        # stream), so a 1-GPU box exercises the real async-collective / stream-ordering code with the real kernels
        self._gather_min_world = 1 if _collective_at_world_1 else 2

    def shard(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S, H, D) -> this rank's heads (a view)."""
        assert x.shape[2] == self.num_heads
        return x[:, :, self.h0:self.h1]

    def q_windows(self, q: torch.Tensor) -> List[Tuple[int, int]]:
        """The q-tile windows ``__call__`` uses for this query shape when ``overlap_windows`` > 1."""
        if self._q_tile_rows is not None:              # test seam (CPU stand-in): no kernel, no device
            bm, slots = self._q_tile_rows, self._slots if self._slots is not None else 256
        else:
            from .flash_attn_interface import device_slots, get_tile_sizes, q_tiles_per_item
            bm, _ = get_tile_sizes(q.shape[-1], q.element_size())
            # the library knows the device and the kernel it would run; both calls map the head dim / element size the same way
            # (80 -> the 96 kernel, e4m3 144 -> its 192 body: ADVICE r4)
            cus, per_cu = device_slots(q.shape[-1], q.element_size())
            slots = cus * per_cu
            unit = q_tiles_per_item(q.shape[-1], q.element_size())
            if unit > 1:       # LA_FLAG_HALF_VOTE: a workgroup item is `unit` q-tiles; windows are planned in items and hold whole items
                q_tiles = -(-q.shape[1] // bm)
                items = plan_q_windows(-(-q_tiles // unit), q.shape[0] * q.shape[2], self.overlap_windows, slots)
                return [(b0 * unit, min(c0 * unit, q_tiles - b0 * unit)) for b0, c0 in items]
        return plan_q_windows(-(-q.shape[1] // bm), q.shape[0] * q.shape[2], self.overlap_windows, slots)

    def preflight_overlapped(self, q, k, v, scale: Optional[float] = None, **kw) -> None:
        """The windowed launches of the overlapped form WITHOUT their collectives: what a rank can check on its own before all ranks
        commit to the overlapped step (a rank that fails once its peers are inside a collective leaves them waiting there). Advances
        the skip state by one call, like any call."""
        if self._windowed is None:
            raise RuntimeError("no windowed attention available (attention_fn stand-in without windowed_attention_fn)")
        self._windowed(q, k, v, self.q_windows(q), lambda i, out, r0, r1: None, scale, **kw)

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
        if self.world >= self._gather_min_world and self.group is not None and gather and self.overlap_windows > 1 \
                and self._windowed is not None:
            return self._call_overlapped(q, k, v, scale, _kernel_events, **kw)
        if _kernel_events is not None:
            _kernel_events[0].record()
        out = self._attention(q, k, v, scale, **kw)
        if _kernel_events is not None:
            _kernel_events[1].record()
        if self.world < self._gather_min_world or self.group is None or not gather:
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


class UlyssesLiteAttention:
    """Sequence-sharded in, sequence-sharded out; heads sharded inside (the DeepSpeed-Ulysses scheme, which is what
    Wan2.x multi-GPU inference uses around its attention). Rank r holds rows [r*S/G, (r+1)*S/G) of q, k, v for ALL heads;
    one all-to-all turns that into all rows of heads [r*H/G, (r+1)*H/G), the QK-Skip attention runs there on this rank's
    own skip state (as in ``HeadShardedLiteAttention``), and a second all-to-all returns the rows. Compared with the
    all-gather driver each rank moves (G-1)/G of ONE shard per direction instead of receiving G-1 shards, and the
    output lands where a sequence-parallel transformer block wants it.

    ``__call__(q, k, v)``: (B, S/G, H, D) each -> (B, S/G, H, D). All ranks must hold the same number of rows."""

    def __init__(self, num_heads: int, enable_skipping: bool = True, threshold: float = -10.0, max_batch_size: int = 4,
                 process_group=None, attention_fn: Optional[Callable[..., torch.Tensor]] = None):
        self.inner = HeadShardedLiteAttention(num_heads, enable_skipping, threshold, max_batch_size, process_group,
                                              attention_fn=attention_fn)
        self.group, self.world, self.rank = process_group, self.inner.world, self.inner.rank
        self.num_heads = num_heads

    @property
    def local(self) -> LiteAttention:
        return self.inner.local

    def _all_to_all(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x, group=self.group)
        return out

    def seq_to_head(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S/G, H, D) on every rank -> (B, S, H/G, D): block g of the send buffer = my rows of rank g's heads."""
        G = self.world
        if G == 1:
            return x
        B, Sl, H, D = x.shape
        send = x.reshape(B, Sl, G, H // G, D).permute(2, 0, 1, 3, 4).contiguous()       # (G, B, Sl, Hl, D)
        recv = self._all_to_all(send)                                                    # block g = rows of rank g
        return recv.permute(1, 0, 2, 3, 4).reshape(B, G * Sl, H // G, D)

    def head_to_seq(self, x: torch.Tensor) -> torch.Tensor:
        """(B, S, H/G, D) -> (B, S/G, H, D): the inverse exchange."""
        G = self.world
        if G == 1:
            return x
        B, S, Hl, D = x.shape
        send = x.reshape(B, G, S // G, Hl, D).permute(1, 0, 2, 3, 4).contiguous()        # (G, B, Sl, Hl, D): rows of rank g
        recv = self._all_to_all(send)                                                    # block g = heads of rank g
        return recv.permute(1, 2, 0, 3, 4).reshape(B, S // G, G * Hl, D)

    def __call__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, **kw) -> torch.Tensor:
        assert q.shape[2] == self.num_heads and k.shape[2] == self.num_heads, "pass all heads of the local rows"
        qh, kh, vh = self.seq_to_head(q), self.seq_to_head(k), self.seq_to_head(v)
        out = self.inner(qh, kh, vh, scale, gather=False, **kw)
        return self.head_to_seq(out)

    def reset_skip_state(self):
        self.inner.reset_skip_state()

    def set_threshold(self, threshold: float):
        self.inner.set_threshold(threshold)


class RingSeqParallelLiteAttention:
    """Ring (context-parallel) attention over sequence shards with the reference's ``SeqParallelLiteAttention`` state
    layout: rank r keeps its query rows; the K/V shards travel round the ring (RCCL send/recv to the next rank, receive
    from the previous one, posted BEFORE the attention on the block in hand so the transfer hides under it), every
    (local Q x K/V shard j) pair owns skip state j (hopper/lite_attention.py:322-345: one LiteAttention per split,
    selected by ``split_idx``), and the G partial results are merged by their LSE (``flash_attn_combine`` — the merge the
    reference's README leaves to the caller, README.md:222-250; oracle hopper/tests/test_flash_attn.py:1178-1187).

    ``__call__(q, k, v)``: (B, S/G, H, D) each -> (B, S/G, H, D)."""

    def __init__(self, enable_skipping: bool = True, threshold: float = -10.0, max_batch_size: int = 4, process_group=None,
                 attention_fn: Optional[Callable[..., Tuple[torch.Tensor, torch.Tensor]]] = None,
                 combine_fn: Optional[Callable[..., torch.Tensor]] = None):
        from .lite_attention import SeqParallelLiteAttention
        self.group = process_group
        if process_group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        else:
            self.world, self.rank = 1, 0
        self.states = SeqParallelLiteAttention(self.world, enable_skipping, threshold, max_batch_size)
        # test seams (CPU/gloo): attention_fn(q, k, v, split_idx, scale) -> (out, lse); combine_fn(outs, lses) -> out
        # e4m3 inputs: SeqParallelLiteAttention asks the library for the reference's row sums when the LSE is returned
        # (it clears the two fp8 fast-form flags): the partial results are merged by it
        self._attention = attention_fn if attention_fn is not None else (
            lambda q, k, v, j, scale, **kw: self.states(q, k, v, j, scale, return_softmax_lse=True, **kw))
        if combine_fn is None:
            from .flash_attn_interface import flash_attn_combine
            combine_fn = lambda outs, lses: flash_attn_combine(outs, lses, return_lse=False)   # noqa: E731
        self._combine = combine_fn

    def __call__(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, *,
                 q_descale: Optional[torch.Tensor] = None, k_descale: Optional[torch.Tensor] = None,
                 v_descale: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``q/k/v_descale`` (e4m3 inputs; fp32 ``(batch, nheads_k)`` as ``flash_attn_func`` takes them): the K / V descales belong to
        the K / V SHARD and travel round the ring with it; q's stays with the local rows."""
        G, r = self.world, self.rank
        def descales(kd, vd):
            if q_descale is None and kd is None and vd is None:
                return {}
            return dict(q_descale=q_descale, k_descale=kd, v_descale=vd)
        if G == 1:
            out, _ = self._attention(q, k, v, 0, scale, **descales(k_descale, v_descale))
            return out
        import torch.distributed as dist
        nxt, prv = (r + 1) % G, (r - 1) % G
        k_cur, v_cur = k.contiguous(), v.contiguous()
        ds_cur = [None if d is None else d.contiguous() for d in (k_descale, v_descale)]
        outs, lses = [], []
        for step in range(G):
            src = (r - step) % G                       # the rank whose K/V shard is in hand = the skip state to use
            reqs = []
            if step + 1 < G:                           # pass the block on while it is being used (read-only here)
                k_nxt, v_nxt = torch.empty_like(k_cur), torch.empty_like(v_cur)
                ds_nxt = [None if d is None else torch.empty_like(d) for d in ds_cur]
                sends = [k_cur, v_cur] + [d for d in ds_cur if d is not None]
                recvs = [k_nxt, v_nxt] + [d for d in ds_nxt if d is not None]
                ops = [dist.P2POp(dist.isend, t, nxt, group=self.group) for t in sends] + \
                      [dist.P2POp(dist.irecv, t, prv, group=self.group) for t in recvs]
                reqs = dist.batch_isend_irecv(ops)
            out, lse = self._attention(q, k_cur, v_cur, src, scale, **descales(*ds_cur))
            outs.append(out)
            lses.append(lse)
            for req in reqs:
                req.wait()
            if step + 1 < G:
                k_cur, v_cur, ds_cur = k_nxt, v_nxt, ds_nxt
        return self._combine(torch.stack(outs), torch.stack(lses))

    def reset_skip_state(self):
        self.states.reset_skip_state()

    def set_threshold(self, threshold: float):
        self.states.set_threshold(threshold)
