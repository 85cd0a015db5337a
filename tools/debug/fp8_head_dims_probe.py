"""e4m3 at head dims 64 / 96 / 128 / 192 / 256, dense S = 16 384, H = 40 (round 6): the five native bodies (round 6; until then 64 / 96 ran zero-padded on the 128 body and 192 / 256 on the bf16
kernels over up-converted operands - the tables of each stage: profiles/r06_fp8_head_dims.md)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liteattention_amd as L                       # noqa: E402

dev = torch.device("cuda", 0)
S, H = 16384, 40
g = torch.Generator(device=dev).manual_seed(0)
for form in ("reference", "encoded"):
    os.environ.pop("LA_FP8_P", None)
    if form == "encoded":
        os.environ["LA_FP8_P"] = "encoded"
    for D in (64, 96, 128, 192, 256):
        q, k, v = [torch.randn(1, S, H, D, device=dev, generator=g).bfloat16().to(torch.float8_e4m3fn) for _ in range(3)]
        for _ in range(20):
            L.flash_attn_func(q, k, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            L.flash_attn_func(q, k, v)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        print(f"fp8 P form {form:9s} head_dim {D:3d}: {ms:7.3f} ms  {4.0 * H * S * S * D / ms / 1e9:7.1f} TFLOP/s (useful)", flush=True)
        del q, k, v
