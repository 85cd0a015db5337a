import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import liteattention_amd as L
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
H, D = 40, 128
def t(fn, n=100):
    for _ in range(30): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for Sq, Sk in ((75088, 512), (75088, 256), (75088, 1024), (32768, 2048), (512, 512)):
    q = torch.randn(1, Sq, H, D, device=dev, generator=g).bfloat16()
    k = torch.randn(1, Sk, H, D, device=dev, generator=g).bfloat16()
    v = torch.randn(1, Sk, H, D, device=dev, generator=g).bfloat16()
    for name, env in (("x64", None), ("v2", "v2")):
        if env: os.environ["LA_FWD_KERNEL"] = env
        else: os.environ.pop("LA_FWD_KERNEL", None)
        ms = t(lambda: L.flash_attn_func(q, k, v))
        print(f"Sq {Sq} Sk {Sk} {name}: {ms:.4f} ms {4.0*H*Sq*Sk*D/ms/1e9:.0f} TFLOP/s", flush=True)
