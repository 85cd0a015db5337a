#!/usr/bin/env python
"""The reference's text + video recipe (/root/reference/README.md:225-246) at the Wan2.1 shape: text = 512 tokens, video = 75 088,
H = 40, D = 128, bf16 - ms and fraction of the MFMA peak of each of the four calls (t2t, t2v, v2t dense; v2v on an imposed 42 % list) and
of the two merges, by HIP events on the launch stream in steady state. `bench.py` imports `joint_recipe` for its sub-record.
    python tools/joint_recipe_bench.py [--fp8] > gpurun_out/joint_recipe.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.selfcheck import banded_rows, executed_flops, impose_lists     # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0


def _steady(launch, est_ms, warm_ms=150.0, timed_ms=300.0, min_reps=5):
    for _ in range(max(2, int(warm_ms / max(est_ms, 0.05)))):
        launch()
    reps = max(min_reps, int(timed_ms / max(est_ms, 0.05)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); launch(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2], reps


def joint_recipe(L, dev, qkv=None, text_len=512, S=75600, H=40, D=128, sparsity=0.42, fp8=False):
    """fp8: the same recipe on e4m3 inputs (`--fp8` on the command line; fractions of the 5 PF fp8 peak; the partials are bf16 either way)."""
    if qkv is None:
        g = torch.Generator(device=dev).manual_seed(1234)
        qkv = [torch.randn(1, S, H, D, device=dev, generator=g).bfloat16() for _ in range(3)]
    if fp8:
        qkv = [x.to(torch.float8_e4m3fn) for x in qkv]
    es, peak = (1, 5000.0) if fp8 else (2, MFMA_BF16_PEAK_TFLOPS)
    q, k, v = qkv
    video = S - text_len
    qt_, qv = q[:, :text_len], q[:, text_len:]
    kt_, kv = k[:, :text_len], k[:, text_len:]
    vt_, vv = v[:, :text_len], v[:, text_len:]
    bm, bn = L.get_tile_sizes(D, es)
    att = L.LiteAttention(threshold=-10.0, max_batch_size=1)
    att.threshold = float("-inf")                      # the imposed v2v list is a fixed point
    att._get_read_write_lists(qv, kv)
    att._phase = 0
    rows = banded_rows(-(-video // bm), -(-video // bn), bm, bn, sparsity)
    impose_lists(att, rows)
    res = {}

    def dense(qq, kk, vv_):
        att.enable_skip_optimization(False)
        r = att(qq, kk, vv_, return_softmax_lse=True)
        att.enable_skip_optimization(True)
        return r
    calls = {"t2t": (lambda: dense(qt_, kt_, vt_), 4.0 * H * text_len * text_len * D),
             "t2v": (lambda: dense(qt_, kv, vv), 4.0 * H * text_len * video * D),
             "v2t": (lambda: dense(qv, kt_, vt_), 4.0 * H * video * text_len * D),
             "v2v": (lambda: att(qv, kv, vv, return_softmax_lse=True), executed_flops(rows, H, 1, video, video, bm, bn, D))}
    outs = {}
    for name, (fn, flops) in calls.items():
        outs[name] = fn()
        ms, reps = _steady(fn, est_ms=max(0.05, flops / 1.2e12 * 1e3))
        res[name] = {"ms": round(ms, 4), "launches_timed": reps, "tflops": round(flops / ms / 1e9, 1),
                     "frac_of_mfma_peak": round(flops / ms / 1e9 / peak, 4)}
    from liteattention_amd.flash_attn_interface import _num_splits
    res["t2v"]["num_splits"] = _num_splits(1, H, text_len, video, D, es, 0)      # what num_splits = -1 (LiteAttention's dense calls) resolves to
    res["t2t"]["num_splits"] = _num_splits(1, H, text_len, text_len, D, es, 0)
    merges = {"merge_text": (lambda: L.flash_attn_combine([outs["t2t"][0], outs["t2v"][0]], [outs["t2t"][1], outs["t2v"][1]]), text_len),
              "merge_video": (lambda: L.flash_attn_combine([outs["v2t"][0], outs["v2v"][0]], [outs["v2t"][1], outs["v2v"][1]]), video)}
    for name, (fn, rows_) in merges.items():
        fn()
        ms, reps = _steady(fn, est_ms=0.5)
        moved = rows_ * H * (3 * D * 2 + 3 * 4)            # two bf16 partials in, one bf16 result out, two LSE in, one out
        res[name] = {"ms": round(ms, 4), "launches_timed": reps, "algorithmic_bytes": moved, "gb_per_s": round(moved / ms / 1e6, 1),
                     "frac_of_hbm_peak": round(moved / ms / 1e6 / 8000.0, 4)}
    small = sum(res[n]["ms"] for n in ("t2t", "t2v", "v2t", "merge_text", "merge_video"))
    res["everything_but_v2v_ms"] = round(small, 4)
    res["everything_but_v2v_over_v2v"] = round(small / res["v2v"]["ms"], 4)
    res["what"] = (f"reference README.md:225-246 at text = {text_len}, video = {video}, H = {H}, D = {D} {'e4m3' if fp8 else 'bf16'}: t2t / t2v / v2t dense through "
                   f"LiteAttention with enable_skip_optimization(False) (host-side split-KV where items < workgroup slots), v2v on the imposed "
                   f"{sparsity:.0%} list, merges by flash_attn_combine on the separate partials (la_combine_list); HIP events, steady state, median; tiles {bm}x{bn}")
    return res


if __name__ == "__main__":
    import liteattention_amd as L
    print(json.dumps(joint_recipe(L, torch.device("cuda", 0), fp8="--fp8" in sys.argv)))
