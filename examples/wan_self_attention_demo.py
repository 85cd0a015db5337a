#!/usr/bin/env python
"""A Wan2.x-style self-attention block running on `from lite_attention import LiteAttention` — the integration recipe of the
reference's README (README.md:268-323) applied to a self-contained stand-in for `WanSelfAttention` (Wan2.1
wan/modules/model.py): q/k/v projections, RMSNorm on q and k, 3-axis RoPE over the (frames, height, width) latent grid,
q/k/v cast to bf16, ONE `self.lite_attention(q, k, v)` call, result back to fp32, output projection. Random weights (no
checkpoints here); what it demonstrates is the drop-in at the module boundary: shapes, dtypes, one LiteAttention instance
per layer, the skip state carried across denoising steps.

    python examples/wan_self_attention_demo.py [--frames 5 --height 16 --width 16 --heads 4 --steps 6 --threshold -6]

Prints, per denoising step, the fraction of tiles skipped and the error against the same block with skipping disabled.
"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lite_attention import LiteAttention  # noqa: E402  (same import line as the reference recipe)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps, self.weight = eps, nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x) * self.weight


def rope_3d(x, grid, theta=10000.0):
    """x (B, S, H, D) with S = F*Hh*W tokens in (f, h, w) order; the head dim is split over the three axes as Wan does
    (d - 2*(d//3), d//3, d//3), each part rotated by its own coordinate."""
    B, S, H, D = x.shape
    F, Hh, W = grid
    parts = [D - 2 * (D // 3), D // 3, D // 3]
    parts = [p - (p % 2) for p in parts]
    parts[0] = D - parts[1] - parts[2]
    f = torch.arange(F, device=x.device).view(F, 1, 1).expand(F, Hh, W).reshape(-1)
    h = torch.arange(Hh, device=x.device).view(1, Hh, 1).expand(F, Hh, W).reshape(-1)
    w = torch.arange(W, device=x.device).view(1, 1, W).expand(F, Hh, W).reshape(-1)
    out, off = [], 0
    for pos, d in zip((f, h, w), parts):
        freqs = 1.0 / theta ** (torch.arange(0, d, 2, device=x.device).float() / d)
        ang = pos.float()[:, None] * freqs[None]                       # (S, d/2)
        cos, sin = ang.cos()[None, :, None], ang.sin()[None, :, None]
        xs = x[..., off:off + d].float().reshape(B, S, H, d // 2, 2)
        a, b = xs[..., 0], xs[..., 1]
        out.append(torch.stack((a * cos - b * sin, a * sin + b * cos), dim=-1).reshape(B, S, H, d))
        off += d
    return torch.cat(out, dim=-1).type_as(x)


class WanLikeSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, threshold=-10.0, enable_skipping=True, max_batch_size=1):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = RMSNorm(dim), RMSNorm(dim)
        # the recipe: ONE LiteAttention per attention layer (its skip lists are that layer's state)
        self.lite_attention = LiteAttention(enable_skipping=enable_skipping, threshold=threshold, max_batch_size=max_batch_size)

    def forward(self, x, grid):
        b, s, n, d = *x.shape[:2], self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, s, n, d)
        k = self.norm_k(self.k(x)).view(b, s, n, d)
        v = self.v(x).view(b, s, n, d)
        q_rope, k_rope = rope_3d(q, grid), rope_3d(k, grid)
        x = self.lite_attention(q_rope.bfloat16(), k_rope.bfloat16(), v.bfloat16())       # (B, S, H, D) bf16
        return self.o(x.float().flatten(2))


class WanLikeCrossAttention(nn.Module):
    """The text-conditioning attention of a Wan2.x block as the stock pipeline writes it: its `flash_attention()` wrapper imports
    `flash_attn` / `flash_attn_interface` and calls `flash_attn_varlen_func` on packed (total, H, D) tensors with `cu_seqlens`
    because the prompts of a batch have different lengths. With `compat_shims/` on the path those imports resolve to the gfx950
    kernel: one dense launch for the whole ragged batch, no host sync."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = RMSNorm(dim), RMSNorm(dim)

    def forward(self, x, context, context_lens):
        import flash_attn_interface                                 # FA3 name, served by compat_shims/flash_attn_interface.py
        b, s, n, d = *x.shape[:2], self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b * s, n, d).bfloat16()
        lens = [int(l) for l in context_lens]
        k = torch.cat([self.norm_k(self.k(context[i, :l])) for i, l in enumerate(lens)]).view(-1, n, d).bfloat16()
        v = torch.cat([self.v(context[i, :l]) for i, l in enumerate(lens)]).view(-1, n, d).bfloat16()
        cu_q = torch.arange(0, (b + 1) * s, s, dtype=torch.int32, device=x.device)
        cu_k = torch.tensor([0] + lens, device=x.device).cumsum(0).to(torch.int32)
        out = flash_attn_interface.flash_attn_varlen_func(q, k, v, cu_q, cu_k, s, max(lens))
        return self.o(out.view(b, s, n * d).float())

    def reference(self, x, context, context_lens):
        b, s, n, d = *x.shape[:2], self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, s, n, d).bfloat16().float()
        outs = []
        for i, l in enumerate(int(l) for l in context_lens):
            k = self.norm_k(self.k(context[i, :l])).view(l, n, d).bfloat16().float()
            v = self.v(context[i, :l]).view(l, n, d).bfloat16().float()
            p = torch.softmax(torch.einsum("qhd,khd->hqk", q[i], k) * d ** -0.5, -1)
            outs.append(torch.einsum("hqk,khd->qhd", p, v).reshape(s, n * d))
        return self.o(torch.stack(outs))


def run(frames=5, height=16, width=16, heads=4, steps=6, threshold=-6.0, seed=0, device="cuda", verbose=True):
    torch.manual_seed(seed)
    dim, grid = heads * 128, (frames, height, width)
    S = frames * height * width
    sparse = WanLikeSelfAttention(dim, heads, threshold=threshold).to(device)
    # random q and k projections give unstructured scores (nothing to skip); tie them, as a stand-in for trained weights
    # under which tokens of the same frame attend to each other
    sparse.k.load_state_dict(sparse.q.state_dict())
    dense = WanLikeSelfAttention(dim, heads, enable_skipping=False).to(device)
    dense.load_state_dict(sparse.state_dict())
    # a latent with frame-to-frame structure, denoised over `steps` steps (noise level 0.5 -> 0.05)
    base = torch.randn(1, frames, 1, dim, device=device).expand(1, frames, height * width, dim).reshape(1, S, dim) * 2.0
    base = base + 0.5 * torch.randn(1, S, dim, device=device)
    rows = []
    with torch.no_grad():
        for t in range(steps):
            sigma = 0.5 + (0.05 - 0.5) * t / max(1, steps - 1)
            x = (1 - sigma ** 2) ** 0.5 * base + sigma * torch.randn(1, S, dim, device=device)
            y, y_ref = sparse(x, grid), dense(x, grid)
            skipped = sparse.lite_attention.get_skip_fraction(batch=1)
            err = (y - y_ref).abs().max().item() / y_ref.abs().max().item()
            rows.append((t, skipped, err))
            if verbose:
                print(f"step {t}: tiles skipped by the NEXT step {skipped:6.1%}   max rel. error vs dense block {err:.2e}")
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    for name, default in (("frames", 5), ("height", 16), ("width", 16), ("heads", 4), ("steps", 6)):
        ap.add_argument(f"--{name}", type=int, default=default)
    ap.add_argument("--threshold", type=float, default=-6.0)
    a = ap.parse_args()
    run(a.frames, a.height, a.width, a.heads, a.steps, a.threshold)
