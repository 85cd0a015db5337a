"""GPU box: is the DENSE call (flash_attn_func: no lists, one workgroup per item, static XCD-aware map) slower than the same work issued
with dense LISTS (persistent workgroups + ticket queues)? Same tensors, interleaved, steady state."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import liteattention_amd as L
from bench import banded_rows, impose_lists, steady_state_ms
for S, H in ((75600, 40), (32768, 40), (16384, 40)):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
    bm, bn = L.get_tile_sizes(128, 2)
    att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf"); att(q, k, v)
    impose_lists(att, banded_rows(-(-S // bm), -(-S // bn), bm, bn, 0.0))
    est = 4.0 * H * S * S * 128 / 1.3e12
    res = {}
    for rep in range(2):
        for name, fn in (("flash_attn_func (static)", lambda: L.flash_attn_func(q, k, v)), ("dense lists (tickets)", lambda: att(q, k, v))):
            ms, n = steady_state_ms(fn, est)
            res.setdefault(name, []).append(ms)
    print(f"S={S} H={H}: " + " | ".join(f"{k_}: {min(v_):.3f} ms {4 * H * S * S * 128 / min(v_) / 1e9:.0f} TF" for k_, v_ in res.items()))
    del q, k, v, att
