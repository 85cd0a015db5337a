#!/usr/bin/env python
"""Round 6: the shapes beside the headline ("the shapes nobody has looked at") - dense and list-walking launches over sequence lengths,
batch sizes, head counts, GQA, cross-attention shapes; HIP events in steady state; useful TFLOP/s and fraction of the bf16 MFMA peak. Every
result is also checked for finiteness (a crash or a NaN here is a finding). -> gpurun_out/shape_sweep.json, profiles/r06_shape_sweep.md
`--fp8`: the e4m3 kernels (all five native head dims; fraction of the 5 PF fp8 peak) -> gpurun_out/shape_sweep_fp8.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L                                     # noqa: E402
from tools.selfcheck import banded_rows, executed_flops, impose_lists     # noqa: E402

dev = torch.device("cuda", 0)
FP8 = "--fp8" in sys.argv
PEAK = 5000.0 if FP8 else 2500.0


def steady(fn, est_ms):
    for _ in range(max(3, int(100.0 / max(est_ms, 0.02)))):
        fn()
    reps = max(5, min(400, int(200.0 / max(est_ms, 0.02))))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(B, Sq, Sk, H, Hk, D, sparsity=None, splits=1):
    g = torch.Generator(device=dev).manual_seed(B + Sq + Sk + H)
    q = torch.randn(B, Sq, H, D, device=dev, generator=g).bfloat16()
    k = torch.randn(B, Sk, Hk, D, device=dev, generator=g).bfloat16()
    v = torch.randn(B, Sk, Hk, D, device=dev, generator=g).bfloat16()
    if FP8:
        q, k, v = [x.to(torch.float8_e4m3fn) for x in (q, k, v)]
    flops = 4.0 * B * H * Sq * Sk * D
    if sparsity is None:
        fn = lambda: L.flash_attn_func(q, k, v, num_splits=splits)            # noqa: E731
    else:
        bm, bn = L.get_tile_sizes(D, 1 if FP8 else 2)
        att = L.LiteAttention(threshold=-10.0, max_batch_size=B)
        att.threshold = float("-inf")
        att._get_read_write_lists(q, k)
        att._phase = 0
        rows = banded_rows(-(-Sq // bm), -(-Sk // bn), bm, bn, sparsity)
        impose_lists(att, rows)
        flops = executed_flops(rows, H, B, Sq, Sk, bm, bn, D)
        fn = lambda: att(q, k, v)                                               # noqa: E731
    out = fn()
    ok = bool(torch.isfinite(out.float()).all().item())
    ms = steady(fn, flops / 1.0e12 * 1e3)
    return {"B": B, "Sq": Sq, "Sk": Sk, "H": H, "Hk": Hk, "D": D, "sparsity": sparsity, "num_splits": splits, "ms": round(ms, 4),
            "tflops": round(flops / ms / 1e9, 1), "frac": round(flops / ms / 1e9 / PEAK, 4), "finite": ok}


cases = []
for S in (1024, 2048, 4096, 8192, 16384, 32768):
    cases.append((1, S, S, 40, 40, 128, None, 1))
    cases.append((1, S, S, 40, 40, 128, None, -1))
    cases.append((1, S, S, 40, 40, 128, 0.42, 1))
for B, S in ((2, 16384), (8, 4096), (16, 1024)):
    cases.append((B, S, S, 40, 40, 128, None, 1))
    cases.append((B, S, S, 40, 40, 128, 0.42, 1))
cases += [(1, 16384, 16384, 8, 8, 128, None, 1), (1, 16384, 16384, 8, 8, 128, None, -1), (1, 16384, 16384, 40, 8, 128, None, 1), (1, 16384, 16384, 40, 1, 128, 0.42, 1),
          (1, 75600, 512, 40, 40, 128, None, 1), (1, 512, 75600, 40, 40, 128, None, 1), (1, 512, 75600, 40, 40, 128, None, -1), (2, 4096, 77, 24, 24, 128, None, 1),
          (1, 4096, 4096, 24, 24, 64, None, 1), (1, 4096, 4096, 24, 24, 64, 0.42, 1), (1, 4096, 4096, 16, 16, 256, None, 1), (1, 4096, 4096, 16, 16, 256, None, -1)]
if FP8:
    cases = []
    for S in (2048, 8192, 32768):
        cases += [(1, S, S, 40, 40, 128, None, 1), (1, S, S, 40, 40, 128, 0.42, 1)]
    for D in (64, 96, 192, 256):
        cases += [(1, 8192, 8192, 40, 40, D, None, 1), (1, 8192, 8192, 40, 40, D, 0.42, 1)]
    cases += [(8, 4096, 4096, 40, 40, 128, None, 1), (16, 1024, 1024, 40, 40, 64, 0.42, 1), (1, 16384, 16384, 40, 8, 256, None, 1), (1, 16384, 16384, 40, 1, 96, 0.42, 1),
              (1, 75600, 512, 40, 40, 128, None, 1), (1, 512, 75600, 40, 40, 128, None, 1), (1, 512, 75600, 40, 40, 128, None, -1), (2, 4096, 77, 24, 24, 64, None, 1), (1, 4096, 4096, 16, 16, 80, None, 1),
              (1, 4096, 4096, 16, 16, 160, 0.42, 1)]
res = []
for c in cases:
    try:
        r = run(*c)
    except Exception as e:  # noqa: BLE001
        r = {"case": list(c), "error": repr(e)}
    res.append(r)
    print(json.dumps(r), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "shape_sweep_fp8.json" if FP8 else "shape_sweep.json"), "w"), indent=1)
