#!/usr/bin/env python
"""GPU box: la_blockmask_to_lists (one wave per list row) against the torch tensor-op form it replaced (round 3), at the Wan2.1 list
geometry (B = 1, H = 40, Qt = 296, Kt = 1182: 11 840 rows of 1 183 ints). HBM-bound byte work: algorithmic bytes = mask bytes read once
+ list ints written once; reported as GB/s against the 8 TB/s HBM roof (/opt/skills/guides/MI355X_MICROARCH.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liteattention_amd import compat  # noqa: E402

dev = "cuda"
B, H, Qt, Kt = 1, 40, 296, 1182
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


for name, mask in (("shared mask [Qt, Kt], p = 0.58", torch.rand(Qt, Kt, device=dev, generator=g) < 0.58),
                   ("per-(batch, head) mask [B, H, Qt, Kt], p = 0.58", torch.rand(B, H, Qt, Kt, device=dev, generator=g) < 0.58),
                   ("per-head mask, fragmented (p = 0.5 alternating-ish)", (torch.rand(B, H, Qt, Kt, device=dev, generator=g) < 0.5))):
    mask[..., -1] = True
    ms_k = timed(lambda: compat.blockmask_to_lists(mask, validate=False, batch=B, heads=H))
    full = mask if mask.dim() == 4 else mask[None, None].expand(B, H, Qt, Kt)
    ms_t = timed(lambda: compat._blockmask_to_lists_host(full, validate=False), n=5)
    a = compat.blockmask_to_lists(mask, validate=False, batch=B, heads=H)
    b = compat._blockmask_to_lists_host(full, validate=False)
    byts = mask.numel() + B * H * Qt * (Kt + 1) * 4
    print(f"{name}: kernel {ms_k * 1e3:.1f} us = {byts / ms_k / 1e6:.0f} GB/s algorithmic ({byts / 1e6:.1f} MB; {byts / ms_k / 1e6 / 8000:.3f} of the 8 TB/s roof) | torch tensor ops "
          f"{ms_t * 1e3:.0f} us ({ms_t / ms_k:.0f}x) | identical: {bool(torch.equal(a, b.reshape(a.shape)))}")
