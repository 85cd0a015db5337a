"""GPU box: bf16 head_dim-64 kernel, dense S=32768 H=80 (and an imposed 42 % list)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
from bench import banded_rows, impose_lists, executed_flops
S, H, D = 32768, 80, 64
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
bm, bn = L.get_tile_sizes(D, 2)
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
res = []
for s in (0.0, 0.42):
    rows = banded_rows(-(-S // bm), -(-S // bn), bm, bn, s)
    impose_lists(att, rows)
    fl = executed_flops(rows, H, 1, S, S, bm, bn, D)
    for _ in range(2): att(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 6
    for _ in range(n): att(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    res.append(f"s={s}: {dt * 1e3:.2f} ms {fl / dt / 1e12:.0f} TF")
print("d64", (bm, bn), " | ".join(res))
