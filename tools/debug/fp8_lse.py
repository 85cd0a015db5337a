"""Debug / evidence: error of the fp8 kernel's O and LSE against the oracle (exact row sums) on a few shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import structured_qkv, fragmented_qkv
from oracle import oracle as orc
import liteattention_amd as L
F8 = torch.float8_e4m3fn
bm, bn = L.get_tile_sizes(128, 1)
def run(name, q, k, v):
    q, k, v = [x.to(F8) for x in (q, k, v)]
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    o8, l8, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round="fp8")
    o32, l32, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=False)
    eo, el = (out.float().cpu() - o8).abs().max().item(), (lse.cpu() - l8).abs().max().item()
    eo32 = (out.float().cpu() - o32).abs().max().item()
    print(f"{name:40s} max|O| {o8.abs().max().item():.3f}  |O-oracle_fp8P| {eo:.4f}  |O-oracle_fp32P| {eo32:.4f}  |oracle_fp8P-oracle_fp32P| {(o8-o32).abs().max().item():.4f}  LSE err max {el:.5f} mean {(lse.cpu()-l8).abs().mean().item():.6f}")
g = torch.Generator().manual_seed(0)
for (B, Sq, H, Sk) in [(1, 17, 1, 17), (2, 129, 3, 65), (1, 1000, 2, 1250), (1, 128, 1, 4224), (1, 300, 2, 1), (1, 300, 2, 3), (1, 300, 2, 12)]:
    run(f"randn B{B} Sq{Sq} H{H} Sk{Sk}", torch.randn(B, Sq, H, 128, generator=g), torch.randn(B, Sk, H, 128, generator=g), torch.randn(B, Sk, H, 128, generator=g))
run("structured S1536", *structured_qkv(1, 1536, 2, 128, seed=300, alpha=9.0, dtype=torch.float32))
run("fragmented S8300", *fragmented_qkv(1, 8300, 3, 128, seed=9, dtype=torch.float32))
S = 1536
q, k, v = torch.randn(1, S, 2, 128, generator=g), torch.randn(1, S, 2, 128, generator=g), torch.randn(1, S, 2, 128, generator=g)
k = k * torch.linspace(8.0, 1.0, S).view(1, S, 1, 1)
run("late-growing max gain 8", q, k, v)
