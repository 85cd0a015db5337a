"""GPU box, library built with -DLA_PROFILE_PHASES: per-item stage costs on REAL (fragmented) lists of the 50-step workload."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from liteattention_amd import _cabi
from tools.selfcheck import DenoiseWorkload
lib = _cabi.load()
lib.la_debug_phase_cycles.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
names = ["-", "ticket", "zero flags + expand list", "table + params + first DMA", "asm body (prologue + tiles + epilogue)", "-", "write list + barrier"]
buf = (ctypes.c_ulonglong * 8)()
wl = DenoiseWorkload(40, torch.device("cuda", 0))
for thr in (-4.22, -2.46):
    att = L.LiteAttention(threshold=thr, max_batch_size=1)
    for t in range(wl.steps):
        q, k, v = wl.qkv(t)
        att(q, k, v)
    att.threshold = float("-inf")
    att._skip_list[1 - att._phase].copy_(att._skip_list[att._phase])
    sp = att.get_skip_fraction(batch=1)
    rl = att.current_read_list()[0]
    ranges = (rl[..., 0].float().mean().item()) / 2
    for _ in range(2): att(q, k, v)
    torch.cuda.synchronize(); lib.la_debug_phase_cycles(buf, 1)
    n = 5
    for _ in range(n): att(q, k, v)
    torch.cuda.synchronize(); lib.la_debug_phase_cycles(buf, 1)
    items = buf[7]
    print(f"thr {thr}: sparsity {sp:.3f}, {ranges:.0f} ranges per row on average, {items} items; cycles per item:")
    for i in (1, 2, 3, 4, 6):
        print(f"   {names[i]:42s} {buf[i] / items:10.0f}")
