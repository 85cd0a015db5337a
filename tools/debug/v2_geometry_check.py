import sys, torch
sys.path.insert(0, "/root/repo")
import liteattention_amd as L
from tools.selfcheck import DenoiseWorkload
wl = DenoiseWorkload(4, torch.device("cuda", 0))
print("tiles", L.get_tile_sizes(128, 2))
for thr in (-4.22, -2.46):
    att = L.LiteAttention(threshold=thr, max_batch_size=1)
    for t in range(wl.steps):
        q, k, v = wl.qkv(t); att(q, k, v)
    print(f"kernel tiles {L.get_tile_sizes(128, 2)}, thr {thr}: {100 * att.get_skip_fraction(batch=1):.1f} % after 50 steps")
