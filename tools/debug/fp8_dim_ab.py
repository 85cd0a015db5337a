"""GPU box: same-box A/B of library variants on e4m3 at one head dim (dense S = 16 384, H = 40; the three forms of P), one subprocess per
(variant, repetition), interleaved.   python tools/debug/fp8_dim_ab.py [--dim 64] [--reps 2] name=path/to/lib.so ...   (the in-tree library is "tree")"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
import liteattention_amd as L
S, H, D = 16384, 40, %d
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16().to(torch.float8_e4m3fn) for _ in range(3)]
out = []
for form in ("reference", "mfma_rowsum", "encoded"):
    os.environ.pop("LA_FP8_P", None)
    if form != "reference":
        os.environ["LA_FP8_P"] = form
    for _ in range(20): L.flash_attn_func(q, k, v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(60): L.flash_attn_func(q, k, v)
    e1.record(); torch.cuda.synchronize()
    out.append("%%.4f" %% (e0.elapsed_time(e1) / 60))
print("RESULT " + " ".join(out))
''' % (ROOT, int(sys.argv[sys.argv.index('--dim') + 1]) if '--dim' in sys.argv else 64)
args = [a for a in sys.argv[1:] if "=" in a]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
variants = [("tree", None)] + [tuple(a.split("=", 1)) for a in args]
res = {n: [] for n, _ in variants}
for r in range(reps):
    for name, path in variants:
        env = dict(os.environ)
        env.pop("LITEATTENTION_AMD_LIB", None)
        if path:
            env["LITEATTENTION_AMD_LIB"] = os.path.abspath(path)
        p = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=600)
        line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
        if not line:
            print(name, "FAILED", p.stderr[-400:])
            continue
        res[name].append([float(x) for x in line[0].split()[1:]])
print(f"{'variant':24s} reference  mfma_rowsum  encoded   (ms, dense S=16384 H=40 e4m3 at the head dim of --dim; median of {reps})")
for name, rows in res.items():
    if rows:
        print(f"{name:24s} " + "  ".join(f"{statistics.median(r[i] for r in rows):8.4f}" for i in range(3)))
