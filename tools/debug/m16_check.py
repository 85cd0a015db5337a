"""GPU box: quick look at where a body under development differs from an fp32 torch attention (dense): error per 16-row block and
per 16-column block of O, and the LSE, for a few (Sq, Sk) shapes. LITEATTENTION_AMD_LIB selects the variant."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import liteattention_amd as L
torch.manual_seed(0)
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(64, 64), (129, 65), (256, 128), (256, 192), (300, 1000), (1000, 1000)]
for Sq, Sk in shapes:
    g = torch.Generator().manual_seed(Sq + Sk)
    q = torch.randn(1, Sq, 2, 128, generator=g).bfloat16().cuda()
    k = torch.randn(1, Sk, 2, 128, generator=g).bfloat16().cuda()
    v = torch.randn(1, Sk, 2, 128, generator=g).bfloat16().cuda()
    out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * 128 ** -0.5
    ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v.float())
    lref = torch.logsumexp(s, -1)
    e = (out.float() - ref).abs()[0, :, 0]                       # head 0: [Sq, D]
    el = (lse - lref).abs()[0, 0]
    print(f"Sq={Sq} Sk={Sk}: max|O-ref| = {e.max().item():.4f} (head 1: {(out.float() - ref).abs()[0, :, 1].max().item():.4f})  max|LSE-ref| = {el.max().item():.5f}")
    rows = e.amax(1)
    rb = [f"{rows[i:i + 16].max().item():.3f}" for i in range(0, min(Sq, 256), 16)]
    cb = [f"{e[:, i:i + 16].max().item():.3f}" for i in range(0, 128, 16)]
    print("   per 16-row block :", " ".join(rb))
    print("   per 16-col block :", " ".join(cb))
    lb = [f"{el[i:i + 16].max().item():.4f}" for i in range(0, min(Sq, 256), 16)]
    print("   LSE per row block:", " ".join(lb))
