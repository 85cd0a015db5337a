"""GPU box: bf16 head_dim-256 kernel (functional instantiation), dense S=16384 H=40."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
for D, S, H in ((256, 16384, 40), (192, 16384, 40), (96, 16384, 40)):
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    for _ in range(2): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 5
    for _ in range(n): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"D={D}: dense S={S} H={H}: {dt * 1e3:.2f} ms {4 * H * S * S * D / dt / 1e12:.0f} TF (useful FLOPs at D={D})")
