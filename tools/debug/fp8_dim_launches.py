"""A handful of dense e4m3 launches at one head dim (LA_PROBE_DIM; S = 16 384, H = 40) for a profiler to look at (tools/fp8_dims_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import liteattention_amd as L                       # noqa: E402

D = int(os.environ.get("LA_PROBE_DIM", "128"))
S, H = 16384, 40
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = [torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16().to(torch.float8_e4m3fn) for _ in range(3)]
for _ in range(12):
    L.flash_attn_func(q, k, v)
torch.cuda.synchronize()
print("done", D)
