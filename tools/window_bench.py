"""GPU box, 1 GPU: what splitting one attention into q-tile windows costs on the kernel side (no collective here).
Shapes = one rank's share at G GPUs of the headline workload (H = 40 / G heads, S = 75600, 42 % banded lists)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from liteattention_amd.parallel import plan_q_windows
from bench import banded_rows, impose_lists

S, D = 75600, 128
bm, bn = L.get_tile_sizes(D, 2)
Qt, Kt = -(-S // bm), -(-S // bn)
for G in (8, 4, 2):
    Hl = 40 // G
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, Hl, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
    att(q, k, v)
    impose_lists(att, banded_rows(Qt, Kt, bm, bn, 0.42))
    res = []
    for n in (1, 2, 3, 4, 6):
        w = plan_q_windows(Qt, Hl, n)
        for _ in range(2): att.call_windowed(q, k, v, w)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(6): att.call_windowed(q, k, v, w)
        torch.cuda.synchronize(); res.append(f"n={n} ({len(w)} launches): {(time.perf_counter() - t) / 6 * 1e3:.2f} ms")
    print(f"G={G} Hl={Hl}: " + " | ".join(res))
    del q, k, v
