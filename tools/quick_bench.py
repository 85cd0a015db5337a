"""Quick kernel A/B numbers (GPU box): dense C2, dense C3, imposed 42 % C3. Env LA_FWD_KERNEL / LITEATTENTION_AMD_LIB select the variant."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from bench import banded_rows, impose_lists, executed_flops

def timeit(fn, n=6, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

tag = os.environ.get("LA_FWD_KERNEL", "v2") + ":" + os.path.basename(os.environ.get("LITEATTENTION_AMD_LIB", "default"))
out = []
for S, H in [(32768, 40), (75600, 40)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
    dt = timeit(lambda: L.flash_attn_func(q, k, v))
    out.append(f"dense S={S}: {dt*1e3:.2f} ms {4*H*S*S*128/dt/1e12:.0f} TF")
    if S == 75600:
        att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
        att(q, k, v)
        for s in (0.42, 0.77):
            bm, bn = L.get_tile_sizes(128, 2)
            rows = banded_rows(-(-S // bm), -(-S // bn), bm, bn, s)
            impose_lists(att, rows)
            fl = executed_flops(rows, H, 1, S, S, bm, bn, 128)
            dt = timeit(lambda: att(q, k, v))
            out.append(f"sparse{int(s*100)} S={S}: {dt*1e3:.2f} ms exec {fl/dt/1e12:.0f} TF")
    del q, k, v
print(tag, " | ".join(out))
