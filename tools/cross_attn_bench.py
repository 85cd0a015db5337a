"""GPU box: dense cross-attention shapes of a Wan2.x block (long video queries x a short text / image context) on the same kernel:
per-item fixed costs dominate there (8 key tiles per item at Sk = 512). usage: python tools/cross_attn_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
H, D = 40, 128
g = torch.Generator(device="cuda").manual_seed(0)
for Sq, Sk in [(75600, 512), (75600, 257), (75600, 1024), (32760, 512), (75600, 4096)]:
    q = torch.randn(1, Sq, H, D, device="cuda", generator=g).bfloat16()
    k = torch.randn(1, Sk, H, D, device="cuda", generator=g).bfloat16()
    v = torch.randn(1, Sk, H, D, device="cuda", generator=g).bfloat16()
    for _ in range(3): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): L.flash_attn_func(q, k, v)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    fl = 4.0 * H * Sq * Sk * D
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :512].transpose(1, 2).float(), k.transpose(1, 2).float(), v.transpose(1, 2).float()).transpose(1, 2)
    err = (L.flash_attn_func(q, k, v)[:, :512].float() - ref).abs().max().item()
    print(f"Sq {Sq} Sk {Sk}: {dt * 1e3:.3f} ms  {fl / dt / 1e12:.0f} TFLOP/s  ({-(-Sk // 64)} key tiles per item)  max err {err:.4f}", flush=True)
