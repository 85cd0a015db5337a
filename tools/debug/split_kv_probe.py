"""Where the time of a split-KV dense call goes (round 6): the t2v shape of the text + video recipe (Sq = 512, Sk = 75 088, H = 40, D = 128) unsplit
and with num_splits = 2 .. 8, by HIP events (steady state); and the same through rocprofv3 --kernel-trace for the per-kernel split."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liteattention_amd as L                       # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
Sq, Sk, H, D = 512, 75088, 40, 128
q = torch.randn(1, Sq, H, D, device=dev, generator=g).bfloat16()
k = torch.randn(1, Sk, H, D, device=dev, generator=g).bfloat16()
v = torch.randn(1, Sk, H, D, device=dev, generator=g).bfloat16()


def t(fn, n=200):
    for _ in range(50):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for ns in (1, 2, 3, 4, 6, 8):
    ms = t(lambda: L.flash_attn_func(q, k, v, num_splits=ns, return_softmax_lse=True))
    print(f"num_splits {ns}: {ms:.4f} ms  ({4.0 * H * Sq * Sk * D / ms / 1e9:.0f} TFLOP/s)", flush=True)
