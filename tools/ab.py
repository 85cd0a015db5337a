"""GPU box: same-box A/B of library variants on the headline shape. One subprocess per (variant, repetition), interleaved
(A B A B ...) so that clock/thermal drift of the box hits all variants alike.

    python tools/ab.py [--fp8] [--reps 2] name=path/to/lib.so ... (the in-tree library is always variant "tree")
Prints per variant the median over repetitions: dense S=75600, imposed 42 % and 77 % (ms and executed TFLOP/s)."""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import liteattention_amd as L
from bench import banded_rows, impose_lists, executed_flops
fp8 = %r
S, H, D = 75600, 40, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
if fp8:
    q, k, v = [x.to(torch.float8_e4m3fn) for x in (q, k, v)]
bm, bn = L.get_tile_sizes(D, 1 if fp8 else 2)
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
out = []
for s in (0.0, 0.42, 0.77):
    rows = banded_rows(-(-S // bm), -(-S // bn), bm, bn, s)
    impose_lists(att, rows)
    fl = executed_flops(rows, H, 1, S, S, bm, bn, D)
    for _ in range(3): att(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 8
    for _ in range(n): att(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    out.append("%%.3f %%.1f" %% (dt * 1e3, fl / dt / 1e12))
print("RESULT " + " ".join(out))
'''
args = [a for a in sys.argv[1:] if not a.startswith("--")]
fp8 = "--fp8" in sys.argv
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 2
args = [a for a in args if not a.isdigit()]
variants = [("tree", None)] + [tuple(a.split("=", 1)) for a in args]
res = {n: [] for n, _ in variants}
for r in range(reps):
    for name, lib in variants:
        env = dict(os.environ)
        if lib:
            env["LITEATTENTION_AMD_LIB"] = os.path.abspath(lib)
        p = subprocess.run([sys.executable, "-c", WORKER % (ROOT, fp8)], capture_output=True, text=True, env=env)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print(name, "FAILED", p.stderr[-400:]); continue
        res[name].append([float(x) for x in line[0].split()[1:]])
for name, rows in res.items():
    if not rows: continue
    med = [statistics.median(c) for c in zip(*rows)]
    print(f"{name:12s} dense {med[0]:7.2f} ms {med[1]:6.0f} TF | 42% {med[2]:7.2f} ms {med[3]:6.0f} TF | 77% {med[4]:7.2f} ms {med[5]:6.0f} TF   (n={len(rows)})")
