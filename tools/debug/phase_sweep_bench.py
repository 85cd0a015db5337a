"""GPU box: every generated body at one code placement (library variants built with align:5,pad4:N for ALL bodies: the loop head at
4 N bytes past a 32-byte boundary), steady state: head dims 64 / 96 / 128 / 192 / 256 dense S = 16 384, fp8 three forms at the headline
42 % list. usage: phase_sweep_bench.py name=lib.so ...  (interleaved, 2 reps)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = r'''
import os, sys, torch
sys.path.insert(0, %r)
import liteattention_amd as L
from bench import steady_state_ms, banded_rows, impose_lists
out = []
for D in (64, 96, 128, 192, 256):
    S, H = 16384, 40
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    ms, n = steady_state_ms(lambda: L.flash_attn_func(q, k, v), 4.0 * H * S * S * D / 1.2e12)
    out.append("d%%d:%%.0f" %% (D, 4 * H * S * S * D / ms / 1e9))
    del q, k, v
S, H, D = 75600, 40, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16().to(torch.float8_e4m3fn) for _ in range(3)]
for form, env in (("fp8", {"LA_FP8_P": "encoded"}), ("fp8exp", {"LA_FP8_P": "mfma_rowsum"}), ("fp8exact", {})):
    os.environ.pop("LA_FP8_P", None); os.environ.update(env)
    att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf"); att(q, k, v)
    rows = banded_rows(-(-S // 256), -(-S // 64), 256, 64, 0.42); impose_lists(att, rows)
    ms, n = steady_state_ms(lambda: att(q, k, v), 27.0, timed_ms=400.0)
    out.append("%%s:%%.2fms" %% (form, ms))
print("RESULT " + " ".join(out))
''' % ROOT
variants = [tuple(a.split("=", 1)) for a in sys.argv[1:]]
for rep in range(2):
    for name, lib in variants:
        env = dict(os.environ)
        if lib != "tree":
            env["LITEATTENTION_AMD_LIB"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, "-c", W], capture_output=True, text=True, env=env)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(f"{name:8s} rep {rep}: " + (line[0][7:] if line else "FAILED " + p.stderr[-300:]), flush=True)
