"""Test seam for ``bench.py`` (``LA_BENCH_STANDIN=tests.bench_standin``): a torch restatement of the attention op that runs on the
CPU and honours the read lists, so that bench.py's REAL ``main()`` — argument parsing, the self-launcher, per-rank seeding
(1234 + rank), the head partition (``Hl = H // world``), the agreement on the overlapped form, the timed loop, the JSON merge and
the exit codes — runs end to end with more than one gloo rank where there is no GPU.
It is test infrastructure: the product has no CPU path, and the line bench.py prints through it says so in ``data``.

``LA_BENCH_STANDIN_FAIL_RANK=r`` makes rank r raise inside its first timed step (exit-code test)."""
import os

import torch

from tools.selfcheck import lists_to_bitmap


class StandIn:
    def __init__(self):
        self.att = None
        self.calls = 0
        self.fail_rank = int(os.environ.get("LA_BENCH_STANDIN_FAIL_RANK", "-1"))

    # what bench.py passes on to HeadShardedLiteAttention (its documented seams: attention_fn, windowed_attention_fn, q_tile_rows)
    def attention_kwargs(self):
        return dict(attention_fn=self.attention, windowed_attention_fn=self.windowed, q_tile_rows=256, slots=4)

    def bind(self, att, bm, bn):
        self.att, self.bm, self.bn = att, bm, bn

    def _masked(self, q, k, v, scale=None):
        """fp32 attention over exactly the keys each q-tile's READ list names; advances the ping-pong phase like the op does."""
        la = self.att.local
        rd, _ = la._get_read_write_lists(q, k)                                   # [B, H, Qt, Kt+1]; flips the phase
        B, S, H, D = q.shape
        keep = lists_to_bitmap(rd[:B])                                            # [B, H, Qt, Kt]
        keep = keep.repeat_interleave(self.bm, dim=2)[:, :, :S].repeat_interleave(self.bn, dim=3)[..., : k.shape[1]]
        s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * (D ** -0.5 if scale is None else scale)
        s = s.masked_fill(~keep, float("-inf"))
        lse = torch.logsumexp(s, dim=-1)
        out = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, dim=-1), v.float()).to(q.dtype)
        return out, lse

    def attention(self, q, k, v, scale=None, **kw):
        self.calls += 1
        if self.att is not None and self.att.rank == self.fail_rank and self.calls > 2:
            raise RuntimeError("injected: this rank fails inside the timed loop")
        return self._masked(q, k, v, scale)[0]

    def windowed(self, q, k, v, windows, hook, scale=None, **kw):
        out = self.attention(q, k, v, scale)
        for i, (t0, n) in enumerate(windows):                                     # rows become final window by window
            hook(i, out, t0 * self.bm, min(q.shape[1], (t0 + n) * self.bm))
        return out

    def local_call(self, q, k, v):
        return self._masked(q, k, v)
