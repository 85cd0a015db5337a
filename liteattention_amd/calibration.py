"""Threshold calibration for QK-Skip. The reference exposes only ``set_threshold`` and mentions
"error calibration" (/root/reference/README.md:14); this helper finds the threshold that reaches a target skip
fraction on a given sequence of attention inputs (e.g. the denoising steps of one layer)."""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch

from .lite_attention import LiteAttention


def run_steps(thr: float, qkv_at: Callable[[int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], n_steps: int,
              max_batch_size: int = 1) -> Tuple[List[float], LiteAttention]:
    """Run ``n_steps`` calls at threshold ``thr``; returns the skip fraction of the READ list of every step."""
    att = LiteAttention(threshold=-1.0, max_batch_size=max_batch_size)
    att.threshold = thr
    trace = []
    for t in range(n_steps):
        q, k, v = qkv_at(t)
        trace.append(att.get_skip_fraction(batch=q.shape[0]))
        att(q, k, v)
    trace.append(att.get_skip_fraction())
    return trace, att


def calibrate_threshold(qkv_at, n_steps: int, target_skip: float, lo: float = -20.0, hi: float = -1e-3,
                        iters: int = 10, tol: float = 0.01, max_batch_size: int = 1):
    """Bisection on thr in [lo, hi) (constant over steps) so that the skip fraction of the list the LAST step
    reads hits ``target_skip`` +- tol. Skip fraction is monotone non-decreasing in thr. Returns (thr, trace)."""
    best = None
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        trace, _ = run_steps(mid, qkv_at, n_steps, max_batch_size)
        got = trace[-2]
        if best is None or abs(got - target_skip) < abs(best[2] - target_skip):
            best = (mid, trace, got)
        if abs(got - target_skip) <= tol:
            break
        if got < target_skip:
            lo = mid
        else:
            hi = mid
    return best[0], best[1]
