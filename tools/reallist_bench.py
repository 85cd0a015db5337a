"""Times the kernel on a REAL (fragmented) skip list as a fixed point (thr=-inf), so variants can be A/B'd on
identical work. The list is produced by N denoise steps of tools/denoise_bench's generator with the default lib,
saved to gpurun_out/reallist.pt, then re-used."""
import os, sys, time, subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
path = os.path.join(ROOT, "gpurun_out", "reallist.pt")
dev = torch.device("cuda", 0)
H, S = 40, 75600
if not os.path.exists(path):
    sys.argv = ["x", "--alpha", "6", "--sink-gain", "0.5", "--targets", "0.42", "--iters", "1", "--calib-heads", "1", "--tag", "tmp"]
    os.environ["LA_THR_LO"], os.environ["LA_THR_HI"] = "-4.5", "-4.4"
    import runpy
    ns = runpy.run_path(os.path.join(ROOT, "tools", "denoise_bench.py"), run_name="__main__")
    att = ns["last_att"]
    torch.save({"list": att.current_read_list().cpu(), "q": None}, path)
    print("saved list, skip fraction", att.get_skip_fraction(batch=1))
    sys.exit(0)
d = torch.load(path)
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = [torch.randn(1, S, H, 128, device=dev, generator=g).bfloat16() for _ in range(3)]
att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
att(q, k, v)
att._skip_list[0].copy_(d["list"].to(dev)); att._skip_list[1].copy_(d["list"].to(dev))
frac = att.get_skip_fraction(batch=1)
for _ in range(3): att(q, k, v)
torch.cuda.synchronize(); t = time.perf_counter(); n = 10
for _ in range(n): att(q, k, v)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
dense = L.LiteAttention(enable_skipping=False)
for _ in range(2): dense(q, k, v)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): dense(q, k, v)
torch.cuda.synchronize(); dd = (time.perf_counter() - t) / 5
tag = os.environ.get("LA_FWD_KERNEL", "v2") + ":" + os.path.basename(os.environ.get("LITEATTENTION_AMD_LIB", "default"))
print(f"{tag}: real list skip={frac:.3f}: {dt*1e3:.2f} ms ; dense {dd*1e3:.2f} ms ; ratio {dt/dd:.3f} (ideal {1-frac:.3f})")
