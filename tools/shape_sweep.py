"""Dense TF/s vs sequence length at ~constant grid size (GPU box): separates cache-resident from streaming behaviour."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
for S, H in [(2048, 640), (4096, 320), (8192, 160), (16384, 80), (32768, 40), (75600, 16)]:
    q, k, v = [torch.randn(1, S, H, 128, device="cuda").bfloat16() for _ in range(3)]
    for _ in range(2): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 5
    for _ in range(n): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"S={S} H={H}: {dt*1e3:.2f} ms {4*H*S*S*128/dt/1e12:.0f} TF  (K+V per head {S*512/1e6:.1f} MB)")
