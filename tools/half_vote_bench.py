#!/usr/bin/env python
"""LA_FLAG_HALF_VOTE against the 256-row vote, same box, one process (round 6): (1) the imposed-list sweep at the headline shape in both
list geometries (must stay within 1 % of each other: the form may not tax a launch whose halves agree); (2) the 50-step run at FIXED
thresholds (tools.selfcheck.DENOISE_THRESHOLDS) in both: total ms, last-step sparsity, error against the dense kernel at the last step.
    python tools/half_vote_bench.py [--steps 50] [--thr 21%,42%,57%,77%] [--no-sweep] > gpurun_out/half_vote.json"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L                                    # noqa: E402
from tools.selfcheck import DENOISE_THRESHOLDS, DenoiseWorkload, banded_rows, executed_flops, impose_lists, listed_tiles_of_rows  # noqa: E402


def ev():
    return torch.cuda.Event(enable_timing=True)


def set_mode(mode):
    if mode == "half":
        os.environ["LA_VOTE"] = "half"
    else:
        os.environ.pop("LA_VOTE", None)


def sweep(dev, sparsities=(0.0, 0.21, 0.42, 0.57, 0.77), S=75600, H=40, D=128, reps=8):
    g = torch.Generator(device=dev).manual_seed(1234)
    q, k, v = [torch.randn(1, S, H, D, device=dev, generator=g).bfloat16() for _ in range(3)]
    out = {}
    for mode in ("tile256", "half", "tile256", "half"):           # interleaved twice: box drift shows as the spread between the two passes
        set_mode(mode)
        bm, bn = L.get_tile_sizes(D, 2)
        qt, kt = -(-S // bm), -(-S // bn)
        att = L.LiteAttention(threshold=-10.0, max_batch_size=1)
        att.threshold = float("-inf")
        att._get_read_write_lists(q, k)
        att._phase = 0
        for s_ in sparsities:
            rows = banded_rows(qt, kt, bm, bn, s_)
            impose_lists(att, rows)
            for _ in range(3):
                att(q, k, v)
            es = [(ev(), ev()) for _ in range(reps)]
            for a, b in es:
                a.record(); att(q, k, v); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in es)[reps // 2]
            fl = executed_flops(rows, H, 1, S, S, bm, bn, D)
            out.setdefault(mode, {}).setdefault(str(s_), []).append({"ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1),
                                                                   "listed": round(listed_tiles_of_rows(rows) / (qt * kt), 4)})
    return out


def denoise(dev, names, steps):
    wl = DenoiseWorkload(40, dev)
    res = {}
    thr_of = dict(DENOISE_THRESHOLDS)
    for name in names:
        thr = thr_of[name]
        for mode in ("tile256", "half"):
            set_mode(mode)
            att = L.LiteAttention(threshold=thr, max_batch_size=1)
            ms = []
            for t in range(steps):
                q, k, v = wl.qkv(t)
                if t == steps - 1:
                    sp = att.get_skip_fraction(batch=1)
                a, b = ev(), ev()
                a.record(); out = att(q, k, v); b.record(); torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            ref = L.flash_attn_func(q, k, v)
            d = (out.float() - ref.float()).abs()
            res.setdefault(name, {})[mode] = {"thr": thr, "total_ms": round(sum(ms), 1), "ms_last": round(ms[-1], 3), "sparsity_last_read": round(sp, 4),
                                              "mean_abs_err_vs_dense": float(f"{d.mean().item():.3e}"), "max_abs_err_vs_dense": float(f"{d.max().item():.3e}"),
                                              "tiles": list(L.get_tile_sizes(128, 2))}
            del att, out, ref, d
        r = res[name]
        r["total_ms_half_over_tile256"] = round(r["half"]["total_ms"] / r["tile256"]["total_ms"], 4)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--thr", default="42%,77%")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-denoise", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    res = {}
    if not a.no_sweep:
        res["imposed_sweep"] = sweep(dev)
    if not a.no_denoise:
        res["denoise"] = denoise(dev, [x for x in a.thr.split(",") if x], a.steps)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
