#!/usr/bin/env python
"""GPU box: per library variant (tools/asm_variants.py), one subprocess each, interleaved over `--reps` rounds: the dense S=16384 H=80
bf16 d128 launch of tools/abl_bench.py in a loop for ~3 s with rocm-smi sampled in the middle. Prints and writes, per variant:
ms per launch, TFLOP/s (dense FLOPs: ablations do less real work, the number prices them), socket W, sclk.

    python tools/variant_power.py out.json name=path/lib.so ...      ("tree" = the in-tree library is always first)"""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import liteattention_amd as L
from bench import power_sample
S, H = 16384, 80
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
for _ in range(3): L.flash_attn_func(q, k, v)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in ev:
    a.record(); L.flash_attn_func(q, k, v); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[10]
pw = power_sample(lambda: L.flash_attn_func(q, k, v), int(2500 / ms), 0) or {}
print("RESULT %%.4f %%.1f %%s %%s" %% (ms, 4 * H * S * S * 128 / ms / 1e9, pw.get("socket_w"), pw.get("sclk_mhz")))
'''
out_path = sys.argv[1]
reps = 2
variants = [("tree", None)] + [tuple(a.split("=", 1)) for a in sys.argv[2:]]
res = {n: [] for n, _ in variants}
for r in range(reps):
    for name, lib in variants:
        env = dict(os.environ)
        if lib:
            env["LITEATTENTION_AMD_LIB"] = os.path.abspath(lib)
        p = subprocess.run([sys.executable, "-c", WORKER % ROOT], capture_output=True, text=True, env=env)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")]
        if not line:
            print(name, "FAILED", p.stderr[-300:], flush=True)
            continue
        t = line[0].split()
        res[name].append((float(t[1]), float(t[2]), float(t[3]) if t[3] != "None" else None, float(t[4]) if t[4] != "None" else None))
rows = []
for name, rr in res.items():
    if not rr:
        continue
    med = lambda i: statistics.median(x[i] for x in rr if x[i] is not None) if any(x[i] is not None for x in rr) else None   # noqa: E731
    rows.append({"variant": name, "ms": round(med(0), 3), "tflops_dense_equiv": round(med(1), 1), "socket_w": med(2), "sclk_mhz": med(3), "n": len(rr)})
    print(rows[-1], flush=True)
with open(out_path, "w") as f:
    json.dump({"what": "dense S=16384 H=80 bf16 d128, median of 20 launches by HIP events, rocm-smi while ~2.5 s of queued launches run; "
                       f"{reps} interleaved rounds of one subprocess per variant", "rows": rows}, f, indent=1)
