import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import liteattention_amd as L
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
D = 128
def t(fn, n=300):
    for _ in range(50): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B, S, H in ((1, 1024, 40), (1, 2048, 40), (1, 4096, 40), (1, 8192, 40), (16, 1024, 40), (8, 4096, 40), (1, 4096, 8), (2, 2048, 16)):
    q = torch.randn(B, S, H, D, device=dev, generator=g).bfloat16()
    k = torch.randn(B, S, H, D, device=dev, generator=g).bfloat16()
    v = torch.randn(B, S, H, D, device=dev, generator=g).bfloat16()
    for name, env in (("x64", None), ("v2", "v2")):
        if env: os.environ["LA_FWD_KERNEL"] = env
        else: os.environ.pop("LA_FWD_KERNEL", None)
        ms = t(lambda: L.flash_attn_func(q, k, v))
        print(f"B{B} S{S} H{H} {name}: {ms:.4f} ms {4.0*B*H*S*S*D/ms/1e9:.0f} TFLOP/s", flush=True)
