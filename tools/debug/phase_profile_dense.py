"""GPU box, library built with -DLA_PROFILE_PHASES (LITEATTENTION_AMD_LIB=build_variants/phases.so): per-item stage costs of the
head_dim-128 kernel on DENSE and imposed-list launches of several sequence lengths - what a (batch, head, q-tile) item costs
outside its tiles. Cycles are s_memtime ticks of thread 0 (shader clock); per-tile cost = slope of the body over the tile count."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from liteattention_amd import _cabi
from bench import banded_rows, impose_lists
lib = _cabi.load()
lib.la_debug_phase_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
names = ["-", "ticket / work id", "zero flags + expand list", "table + params + first DMA", "asm body", "-", "write list + barrier"]
buf = (ctypes.c_ulonglong * 8)()
res = {}
for S, H, sp in ((16384, 40, None), (32768, 40, None), (75600, 8, None), (75600, 8, 0.42), (75600, 8, 0.77)):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
    bm, bn = L.get_tile_sizes(128, 2)
    Qt, Kt = -(-S // bm), -(-S // bn)
    if sp is None:
        run = lambda: L.flash_attn_func(q, k, v)
        tiles = Kt
    else:
        att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf"); att(q, k, v)
        rows = banded_rows(Qt, Kt, bm, bn, sp); impose_lists(att, rows)
        run = lambda: att(q, k, v)
        tiles = sum((r[1] - r[2] + 1) + ((r[3] - r[4] + 1) if r[0] == 4 else 0) for r in rows.tolist()) / Qt
    for _ in range(2): run()
    torch.cuda.synchronize(); lib.la_debug_phase_cycles(buf, 1)
    for _ in range(3): run()
    torch.cuda.synchronize(); lib.la_debug_phase_cycles(buf, 1)
    items = buf[7]
    st = {names[i]: buf[i] / items for i in (1, 2, 3, 4, 6)}
    res[(S, sp)] = (tiles, st)
    print(f"S={S} H={H} {'dense' if sp is None else f'imposed {sp}'}: {tiles:.0f} tiles per item, {items} items; cycles per item: " +
          ", ".join(f"{k_}: {v_:.0f}" for k_, v_ in st.items()))
(t0, s0), (t1, s1) = res[(16384, None)], res[(75600, None)]
slope = (s1["asm body"] - s0["asm body"]) / (t1 - t0)
print(f"per tile {slope:.0f} cycles; asm body fixed part (prologue + epilogue) {s0['asm body'] - slope * t0:.0f} cycles; "
      f"shell stages {sum(v_ for k_, v_ in s0.items() if k_ != 'asm body'):.0f} cycles per item")
