import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import liteattention_amd as L
S, H = 16384, 80
q, k, v = [torch.randn(1, S, H, 128, device="cuda").bfloat16() for _ in range(3)]
for _ in range(2): L.flash_attn_func(q, k, v)
torch.cuda.synchronize(); t = time.perf_counter(); n = 8
for _ in range(n): L.flash_attn_func(q, k, v)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
print(os.path.basename(os.environ.get("LITEATTENTION_AMD_LIB", "default")), f"{dt*1e3:.2f} ms {4*H*S*S*128/dt/1e12:.0f} TF")
