"""GPU box: first contact with a new fp8 kernel: tiny dense cases against torch fp32, under the caller's `timeout`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
F8 = torch.float8_e4m3fn
print("tiles", L.get_tile_sizes(128, 1), flush=True)
for (B, Sq, Sk, H) in [(1, 256, 64, 1), (1, 256, 128, 1), (1, 256, 256, 1), (1, 300, 333, 2), (2, 1000, 1250, 3)]:
    g = torch.Generator().manual_seed(Sq + Sk)
    q, k, v = [torch.randn(B, s_, H, 128, generator=g).to(F8) for s_ in (Sq, Sk, Sk)]
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    torch.cuda.synchronize()
    qf, kf, vf = [x.float().transpose(1, 2) for x in (q, k, v)]
    sc = qf @ kf.transpose(-1, -2) / 128 ** 0.5
    ref = (torch.softmax(sc, -1) @ vf).transpose(1, 2)
    lse_ref = torch.logsumexp(sc, -1)
    print((B, Sq, Sk, H), "max|O-ref| %.4f (max|ref| %.2f)  max|lse-ref| %.5f" % ((out.float().cpu() - ref).abs().max().item(), ref.abs().max().item(),
          (lse.cpu() - lse_ref).abs().max().item()), flush=True)
