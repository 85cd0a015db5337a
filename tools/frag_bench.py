"""GPU box: what fragmentation costs. Produces a REAL skip list (50 denoise steps of tools/denoise_bench's generator at a
given threshold), then times the kernel at thr=-inf (fixed point) on (a) that list, (b) a list with the SAME number of
tiles per row as one contiguous range, (c) the same tiles per row but identical for all rows of a head (mean), (d) dense."""
import os, sys, time, runpy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
thr = float(os.environ.get("FRAG_THR", "-2.462"))
sys.argv = ["x", "--alpha", "6", "--sink-gain", "0.5", "--targets", "0.5", "--iters", "1", "--calib-heads", "1", "--tag", "tmp"]
os.environ["LA_THR_LO"], os.environ["LA_THR_HI"] = str(thr - 1e-3), str(thr + 1e-3)
ns = runpy.run_path(os.path.join(ROOT, "tools", "denoise_bench.py"), run_name="__main__")
real = ns["last_att"].current_read_list().clone()          # [1, H, Qt, Kt+1]
del ns
torch.cuda.empty_cache()
dev = real.device
H, Qt, W = real.shape[1:]
Kt = W - 1
body = real.to(torch.int64)
pairs = body[..., 1:1 + 2 * ((W - 1) // 2)].unflatten(-1, (-1, 2))
nr = body[..., 0].clamp_min(2) // 2
live = torch.arange(pairs.shape[-2], device=dev) < nr.unsqueeze(-1)
counts = ((pairs[..., 0] - pairs[..., 1] + 1).clamp_min(0) * live).sum(-1)          # [1, H, Qt]
print(f"real list: sparsity {1 - counts.sum().item() / (H * Qt * Kt):.3f}, ranges/row mean {nr.float().mean().item():.1f} max {nr.max().item()}, "
      f"tiles/row min {counts.min().item()} mean {counts.float().mean().item():.1f} max {counts.max().item()}")

import json
json.dump({"thr": thr, "Kt": Kt, "counts": counts[0].cpu().tolist(), "ranges": nr[0].cpu().tolist()},
          open(os.path.join(ROOT, "gpurun_out", "frag_counts.json"), "w"))


def banded(cnt):
    out = torch.zeros_like(real)
    out[..., 0] = 2
    out[..., 1] = Kt - 1
    out[..., 2] = (Kt - cnt).clamp(0, Kt - 1).to(torch.int32)
    return out

S = 75600
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = [torch.randn(1, S, H, 128, device=dev, generator=g).bfloat16() for _ in range(3)]
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
def timeit(lst):
    att._skip_list[0].copy_(lst); att._skip_list[1].copy_(lst)
    for _ in range(2): att(q, k, v)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 8
    for _ in range(n): att(q, k, v)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
t_real = timeit(real)
t_band = timeit(banded(counts))
t_mean = timeit(banded(counts.float().mean(-1, keepdim=True).round().long().expand_as(counts)))
t_full = timeit(banded(torch.full_like(counts, Kt)))
frac = counts.sum().item() / (H * Qt * Kt)
print(f"sched={os.environ.get('LA_SCHED', 'dynamic')}: real {t_real:.2f} ms | same counts, one range {t_band:.2f} | per-head mean count {t_mean:.2f} | full {t_full:.2f} "
      f"| ideal {t_full * frac:.2f}")
