import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
torch.manual_seed(0)
for (B, Sq, Sk, H) in [(1, 64, 64, 1), (1, 256, 64, 1), (1, 256, 128, 1), (1, 256, 192, 1), (1, 256, 256, 1), (1, 256, 512, 1), (1, 256, 1024, 1), (1, 300, 777, 2), (2, 777, 777, 3)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    q, k, v = [torch.randn(B, S_, H, 128, device="cuda", generator=g).bfloat16() for S_ in (Sq, Sk, Sk)]
    o, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    s = torch.einsum("bthd,bshd->bhts", q.float(), k.float()) * 128 ** -0.5
    ref = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), v.float())
    lref = torch.logsumexp(s, -1)
    e = (o.float() - ref).abs()
    # which rows / d are wrong
    bad = (e > 0.01).nonzero()
    rows = sorted(set(bad[:, 1].tolist()))[:12]
    ds = sorted(set(bad[:, 3].tolist()))[:12]
    print(f"B{B} Sq{Sq} Sk{Sk} H{H}: maxerr {e.max().item():.4f} lse err {(lse - lref).abs().max().item():.4f} nbad {len(bad)} rows {rows} d {ds}")
