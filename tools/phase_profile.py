"""GPU box, library built with -DLA_PROFILE_PHASES (build_variants/phases.so): where a work item's fixed cost goes.
Runs the bf16 x64 kernel on imposed lists of two densities and prints cycles per item for each stage."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from liteattention_amd import _cabi
from bench import banded_rows, impose_lists
lib = _cabi.load()
lib.la_debug_phase_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
S, H, D = 75600, 40, 128
bm, bn = L.get_tile_sizes(D, 2)
Qt, Kt = -(-S // bm), -(-S // bn)
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
names = ["-", "ticket", "zero flags + expand list", "params + first DMA", "asm body (prologue + tiles + epilogue)", "finalize + store O", "write list + barrier"]
buf = (ctypes.c_ulonglong * 8)()
for s in (0.77, 0.95):
    rows = banded_rows(Qt, Kt, bm, bn, s)
    impose_lists(att, rows)
    for _ in range(2): att(q, k, v)
    torch.cuda.synchronize(); lib.la_debug_phase_cycles(buf, 1)
    n = 5
    for _ in range(n): att(q, k, v)
    torch.cuda.synchronize(); lib.la_debug_phase_cycles(buf, 1)
    items = buf[7]
    tiles = sum(r[1] - r[2] + 1 + ((r[3] - r[4] + 1) if r[0] == 4 else 0) for r in rows.tolist()) * H * n
    print(f"sparsity {s}: {items} items, {tiles / items:.1f} tiles/item; cycles per item:")
    for i in range(1, 7):
        print(f"   {names[i]:42s} {buf[i] / items:10.0f}")
    print(f"   total {sum(buf[1:7]) / items:.0f} cycles/item (100 MHz counter? see ratio to wall time)")
