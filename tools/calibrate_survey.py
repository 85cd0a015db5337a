#!/usr/bin/env python
"""The generator SURVEY.md 8(d) pins (tools.selfcheck.DenoiseWorkload(generator="survey")), bisected: constant thresholds in [-20, 0) at
which the list the LAST of 50 denoising steps reads has 21 / 42 / 57 / 77 % (+- 1 %) sparsity, all 40 heads, the kernel's own tile
(missing 5: rounds 4-5 only evaluated it at the anchored generator's thresholds, where it skips nothing, and bisected 42 %).
Per target: thr, the step-49 sparsity, the per-step trace, whether +- 1 % was reached - or the bracket end it ran into.
    python tools/calibrate_survey.py [out.json]      (GPU box; about 4 s per bisection step)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L                                                   # noqa: E402
from liteattention_amd.calibration import calibrate_threshold                   # noqa: E402
from tools.selfcheck import DenoiseWorkload                                     # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "denoise50_survey_calibration.json")
dev = torch.device("cuda", 0)
H, STEPS = 40, 50
res = {"what": "generator='survey' (SURVEY.md 8d): thresholds (log2 units, constant over the 50 steps) for a target sparsity of the step-49 READ "
               f"list; all 40 heads; tiles {L.get_tile_sizes(128, 2)}; bisection in [-20, 0)", "targets": {}}
wl = DenoiseWorkload(H, dev, steps=STEPS, generator="survey")
t0 = time.time()
# the sparsity of this generator moves from 0 to ~100 % between thr = -2 and 0 (a same-frame score is 0.5 nat above a cross-frame one): the
# whole interval [-20, 0) is the bracket of every target, 14 halvings resolve it to 1e-3
for target in (0.21, 0.42, 0.57, 0.77):
    thr, trace = calibrate_threshold(wl.qkv, STEPS, target, lo=-20.0, hi=-1e-3, iters=14, tol=0.004)
    res["targets"][f"{round(target * 100)}%"] = {
        "thr": round(thr, 4), "sparsity_step49_read_list": round(trace[-2], 4), "within_1pct": bool(abs(trace[-2] - target) <= 0.01),
        "mean_sparsity_over_steps": round(sum(trace[:-1]) / STEPS, 4),
        "sparsity_trace_every_5_steps": [round(x, 4) for x in trace[:-1:5]] + [round(trace[-2], 4)], "sparsity_after_step49": round(trace[-1], 4)}
    print(target, res["targets"][f"{round(target * 100)}%"], f"{time.time() - t0:.0f}s", flush=True)
res["seconds"] = round(time.time() - t0, 1)
os.makedirs(os.path.dirname(out_path), exist_ok=True)
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print("wrote", out_path)
