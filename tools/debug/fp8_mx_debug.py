"""GPU box: the fp8 kernel against the oracle on one or two key tiles, error per query row (finds layout / scale mistakes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc
import liteattention_amd as L
F8 = torch.float8_e4m3fn
bm, bn = L.get_tile_sizes(128, 1)
g = torch.Generator().manual_seed(0)
for (Sq, Sk, spread) in [(64, 64, 1.0), (64, 64, 4.0), (64, 128, 4.0), (256, 640, 4.0)]:
    q, k, v = [torch.randn(1, s_, 1, 128, generator=g) for s_ in (Sq, Sk, Sk)]
    k = k * spread
    q, k, v = [x.to(F8) for x in (q, k, v)]
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    ol, ll, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round="fp8_lin")
    o32, l32, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=False)
    e = (out.float().cpu() - o32)[0, :, 0].abs().max(-1).values
    el = (lse.cpu() - l32)[0, 0]
    eo = (ol - o32)[0, :, 0].abs().max(-1).values
    print(f"Sq {Sq} Sk {Sk} spread {spread}: kernel max|O-exact| {e.max():.4f} (oracle fp8_lin: {eo.max():.4f})  LSE err kernel {el.abs().max():.4f} oracle {(ll - l32).abs().max():.4f}")
    bad = (e > 3 * eo.max() + 1e-3).nonzero().flatten().tolist()
    print("   rows off:", bad[:40], "of", Sq, " lse err of first rows:", [round(x, 3) for x in el[:8].tolist()])
