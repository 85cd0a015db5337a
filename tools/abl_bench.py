"""A/B timing of one library variant (GPU box): dense S=16384 H=80 bf16 d128 + a small max-error check vs torch fp32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
g = torch.Generator(device="cuda").manual_seed(1)
qs, ks, vs = [torch.randn(2, 777, 3, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
o = L.flash_attn_func(qs, ks, vs).float()
ref = torch.nn.functional.scaled_dot_product_attention(qs.float().transpose(1, 2), ks.float().transpose(1, 2), vs.float().transpose(1, 2)).transpose(1, 2)
err = (o - ref).abs().max().item()
att = L.LiteAttention(max_batch_size=2, threshold=-3.0)
for _ in range(3): o2 = att(qs, ks, vs)
err2 = (o2.float() - ref).abs().max().item()
S, H = 16384, 80
q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
if os.environ.get("LA_ABL_DTYPE") == "fp8":
    q, k, v = [x.to(torch.float8_e4m3fn) for x in (q, k, v)]
for _ in range(2): L.flash_attn_func(q, k, v)
torch.cuda.synchronize(); t = time.perf_counter(); n = 8
for _ in range(n): L.flash_attn_func(q, k, v)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(os.path.basename(os.environ.get("LITEATTENTION_AMD_LIB", "default")), f"{dt*1e3:.2f} ms {4*H*S*S*128/dt/1e12:.0f} TF  maxerr dense {err:.4f} skip {err2:.4f} skipfrac {att.get_skip_fraction():.2f}")
