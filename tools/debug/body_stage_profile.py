"""GPU box, LITEATTENTION_AMD_LIB=build_variants/stamps.so (LA_X64_OPT=stamps python -m liteattention_amd.build --out=build_variants/stamps.so):
what the FIXED part of the head_dim-128 asm body is made of. The `stamps` body (gen_fwd_x64.py) has wave 0 read the shader clock at six
points of every item and overwrite the first five LSE values of the item's q-tile with the differences; this script collects them per
item on dense and imposed-list launches at the headline sequence length."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from bench import banded_rows, impose_lists
names = ["start -> Q loads issued (parameters, lane constants, addresses)", "Q loads -> Q in registers (wait)",
         "Q in AGPRs, O = 0, K(0) reads, QK(0), K(1) reads, K(2) DMA, first statistics + softmax start", "the loop",
         "epilogue (vote word, 1 / l, O and LSE stores)"]
S, H = 75600, 8
bm, bn = L.get_tile_sizes(128, 2)
Qt, Kt = -(-S // bm), -(-S // bn)
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
for sp in (None, 0.42, 0.77):
    if sp is None:
        run = lambda: L.flash_attn_func(q, k, v, return_softmax_lse=True)
        tiles = float(Kt)
    else:
        att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf"); att(q, k, v)
        rows = banded_rows(Qt, Kt, bm, bn, sp); impose_lists(att, rows)
        run = lambda: att(q, k, v, return_softmax_lse=True)
        tiles = sum((r[1] - r[2] + 1) + ((r[3] - r[4] + 1) if r[0] == 4 else 0) for r in rows.tolist()) / Qt
    for _ in range(3): run()
    _, lse = run()
    torch.cuda.synchronize()
    idx = (torch.arange(Qt, device="cuda") * bm).view(-1, 1) + torch.arange(5, device="cuda").view(1, -1)      # [Qt, 5]
    d = lse[0][:, idx.reshape(-1)].view(H, Qt, 5).reshape(-1, 5).double()                                      # items x 5
    print(f"S={S} H={H} {'dense' if sp is None else f'imposed {sp}'}: {tiles:.0f} tiles per item, {d.shape[0]} items; shader-clock cycles per item (mean / median / p90):")
    for i, n in enumerate(names):
        c = d[:, i]
        print(f"   {c.mean().item():10.0f} {c.median().item():10.0f} {c.quantile(0.9).item():10.0f}   {n}" + (f"   = {c.mean().item() / tiles:.0f} per tile" if i == 3 else ""))
    fixed = d[:, [0, 1, 2, 4]].sum(1)
    print(f"   {fixed.mean().item():10.0f} {fixed.median().item():10.0f} {fixed.quantile(0.9).item():10.0f}   everything but the loop")
