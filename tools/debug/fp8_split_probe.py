import sys, torch
sys.path.insert(0, "/root/repo")
import liteattention_amd as L
F8 = torch.float8_e4m3fn
g = torch.Generator(device="cuda").manual_seed(0)
for D in (128, 64, 256):
    for B in (1, 2):
        q = torch.randn(B, 512, 8, D, device="cuda", generator=g).to(F8)
        k = torch.randn(B, 20000, 8, D, device="cuda", generator=g).to(F8)
        v = torch.randn(B, 20000, 8, D, device="cuda", generator=g).to(F8)
        qd = (0.5 + torch.rand(B, 8, device="cuda", generator=g))
        o1, l1 = L.flash_attn_func(q, k, v, q_descale=qd, k_descale=qd, v_descale=qd, return_softmax_lse=True)
        for ns in (3, -1):
            try:
                o2, l2 = L.flash_attn_func(q, k, v, q_descale=qd, k_descale=qd, v_descale=qd, return_softmax_lse=True, num_splits=ns)
                print(D, B, ns, "O diff", (o1.float() - o2.float()).abs().max().item(), "max|O|", o1.float().abs().max().item(), "LSE diff", (l1 - l2).abs().max().item(), flush=True)
            except Exception as e:
                print(D, B, ns, "ERROR", repr(e)[:300], flush=True)
