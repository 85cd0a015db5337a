"""GPU box: dense head_dim-128 launches of several lengths on library variants of the ticket order (chunk C, heads interleaved G:
-DLA_SCHED_C / -DLA_SCHED_G A/B builds), interleaved, steady state. usage: dense_sched_variants.py name=lib.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = r'''
import os, sys, torch
sys.path.insert(0, %r)
import liteattention_amd as L
from bench import steady_state_ms
out = []
for S, H in ((16384, 40), (32768, 40), (75600, 40)):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
    ms, n = steady_state_ms(lambda: L.flash_attn_func(q, k, v), 4.0 * H * S * S * 128 / 1.3e12)
    out.append("%%d:%%.3f:%%.0f" %% (S, ms, 4 * H * S * S * 128 / ms / 1e9))
    del q, k, v
print("RESULT " + " ".join(out))
''' % ROOT
variants = [("tree", None)] + [tuple(a.split("=", 1)) for a in sys.argv[1:]]
for rep in range(2):
    for name, lib in variants:
        env = dict(os.environ)
        if lib:
            env["LITEATTENTION_AMD_LIB"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, "-c", W], capture_output=True, text=True, env=env)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(f"{name:10s} rep {rep}: " + (line[0][7:] if line else "FAILED " + p.stderr[-300:]), flush=True)
