"""GPU box: ONE configuration of the headline shape (B1 S75600 H40 D128 bf16) for the bytes-vs-sparsity table (north_star: "rocprof
HBM GB/s on skipped tiles"): warm-up, then 3 timed steps; run it under `rocprofv3 --pmc ...` passes (tools/evidence.sh) -
the LAST three forward-kernel dispatches are the probe's.

    --imposed S     banded list of sparsity S (bench.py's lists; thr = -inf)
    --real THR      50 synthetic denoising steps at threshold THR (tools/denoise_bench.py generator with --alpha 6 --sink-gain 0.5, the settings of the committed 50-step runs), then the
                    list reached is frozen (thr = -inf) and timed: real, fragmented, per-head lists
"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from tools import selfcheck as sc

ap = argparse.ArgumentParser()
ap.add_argument("--imposed", type=float, default=None)
ap.add_argument("--real", type=float, default=None)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda", 0)
S, H, D = 75600, 40, 128
bm, bn = L.get_tile_sizes(D, 2)
att = L.LiteAttention(max_batch_size=1)
if a.real is None:
    g = torch.Generator(device=dev).manual_seed(1234)
    q, k, v = [torch.randn(1, S, H, D, device=dev, generator=g).bfloat16() for _ in range(3)]
    att.threshold = float("-inf")
    att(q, k, v)
    sc.impose_lists(att, sc.banded_rows(-(-S // bm), -(-S // bn), bm, bn, a.imposed))
else:
    FR, PER, alpha, rho, sink, sink_gain = 21, 3600, 6.0, 0.85, 640, 0.5            # the generator settings of profiles/r01e_denoise50.json
    g = torch.Generator(device=dev).manual_seed(1234)
    z = torch.randn(FR, H, D, device=dev, generator=g)
    u = torch.empty_like(z); u[0] = z[0]
    for f in range(1, FR):
        u[f] = rho * u[f - 1] + (1 - rho ** 2) ** 0.5 * z[f]
    u = u / u.norm(dim=-1, keepdim=True)
    cen = u[torch.arange(S, device=dev) // PER]
    q0 = alpha * cen + torch.randn(S, H, D, device=dev, generator=g)
    k0 = alpha * cen + torch.randn(S, H, D, device=dev, generator=g)
    anchor = u.mean(0); anchor = anchor / anchor.norm(dim=-1, keepdim=True)
    q0 = q0 + alpha * sink_gain * anchor
    k0[S - sink:] = k0[S - sink:] + alpha * (1 + sink_gain) * anchor
    v0 = torch.randn(S, H, D, device=dev, generator=g)
    base = (q0[None], k0[None], v0[None])
    att.threshold = a.real
    cache = f"/tmp/la_real_lists_{a.real}_{a.steps}_{bm}.pt"      # (per list geometry: LA_VOTE=half has 128-row lists)
    #      # the lists do not depend on the library variant (votes are bit-exact): reuse
    for t in range(a.steps):
        if os.path.exists(cache) and t < a.steps - 1:
            continue
        s_ = 0.5 + (0.05 - 0.5) * t / max(1, a.steps - 1)
        gt = torch.Generator(device=dev).manual_seed(10 ** 6 + t)
        q, k, v = [((1 - s_ * s_) ** 0.5 * x + s_ * torch.randn(x.shape, device=dev, generator=gt)).to(torch.bfloat16) for x in base]
        if os.path.exists(cache):
            att._get_read_write_lists(q, k)
            att._skip_list[att._phase].copy_(torch.load(cache).to(dev))
        else:
            att(q, k, v)
    if not os.path.exists(cache):
        torch.save(att._skip_list[att._phase].cpu(), cache)
    att.threshold = float("-inf")                    # freeze the list reached after `steps` steps
    att._skip_list[1 - att._phase].copy_(att._skip_list[att._phase])
sparsity = att.get_skip_fraction(batch=1)
att(q, k, v)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for i in range(3):
    ev[i].record(); att(q, k, v)
ev[3].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[3]) / 3
listed = (1 - sparsity) * 4.0 * H * S * S * D
print(f"PROBE mode={'real' if a.real is not None else 'imposed'} arg={a.real if a.real is not None else a.imposed} sparsity={sparsity:.4f} "
      f"ms={ms:.3f} executed_tflops={listed / ms / 1e9:.1f}")
