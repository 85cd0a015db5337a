"""GPU box: the bf16 head dims next to the headline 128 - dense S=16384 H=40 at head_dim 256, 192, 96 (or the dims given as arguments).
LA_FWD_KERNEL=v2 runs the hipcc-scheduled A/B kernels instead (192 / 96 are then zero-padded onto 256 / 128 by the host)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
dims = [int(a) for a in sys.argv[1:]] or [256, 192, 96]
for D, S, H in [(d, 16384, 40) for d in dims]:
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    for _ in range(2): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 5
    for _ in range(n): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"D={D}: dense S={S} H={H}: {dt * 1e3:.2f} ms {4 * H * S * S * D / dt / 1e12:.0f} TF (useful FLOPs at D={D})")
