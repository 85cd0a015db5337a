"""GPU box: per-workgroup fixed cost of the x64 kernel. Imposed banded lists of decreasing density at the headline shape;
a linear fit of time against listed tiles separates the per-tile cost from the per-workgroup cost (prologue: list
expansion, Q load, first DMA; epilogue: O store, list write)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from bench import banded_rows, impose_lists, listed_tiles_of_rows

S, H, D = 75600, 40, 128
bm, bn = L.get_tile_sizes(D, 2)
Qt, Kt = -(-S // bm), -(-S // bn)
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
pts = []
for s in (0.0, 0.42, 0.77, 0.9, 0.95, 0.98, 0.995, 0.9995):
    rows = banded_rows(Qt, Kt, bm, bn, s)
    impose_lists(att, rows)
    for _ in range(2): att(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 6
    for _ in range(n): att(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    tiles = listed_tiles_of_rows(rows) * H
    pts.append((tiles, dt))
    print(f"s={s}: listed tiles/WG {tiles / (H * Qt):.1f}  {dt * 1e3:.3f} ms  per listed tile per WG-slot {dt / (tiles / 256) * 1e6:.3f} us")
(t0, d0), (t1, d1) = pts[0], pts[-1]
per_tile = (d0 - d1) / (t0 - t1)
fixed = d1 - per_tile * t1
print(f"per tile {per_tile * 256 * 1e6:.3f} us per CU ; fixed {fixed * 1e3:.3f} ms per call = {fixed / (H * Qt / 256) * 1e6:.2f} us per workgroup")
