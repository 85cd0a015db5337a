#!/usr/bin/env python
"""BASELINE.json configs[2]: 1xMI355X bf16 S=75600 (Wan2.1-14B video shape), QK-Skip at thresholds that yield
21/42/57/77 % sparsity over 50 synthetic denoising steps. Writes profiles/<tag>_denoise50.json.

Synthetic generator (structured, slowly varying — iid randn gives ~0 % sparsity at any negative threshold):
S = 21 frames x 3600 tokens. Per head: frame centroids u_f follow an AR(1) walk (corr(u_f,u_g) = rho^|f-g|), so
scores decay smoothly with frame distance; q0 = a*u_f + n, k0 = a*u_f + n, v0 = n. QK-Skip walks key tiles in
DESCENDING order and compares a tile with the running max BEFORE it, so only tiles met after the row's dominant
keys can be flagged; real video attention has global anchor tokens, modelled here by making the LAST `sink`
tokens keys that every query scores highly (a*a*sink_gain). Step t: x_t = sqrt(1-s_t^2) x0 + s_t n_t with s_t
linear 0.5 -> 0.05 (noise seeds 10^6+t). Thresholds: bisection on thr in [-24, 0) (constant over steps) so that the
sparsity of the list read by the LAST step hits the target; calibrated on `--calib-heads` heads, then run on all 40.
"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="r01")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--heads", type=int, default=40)
ap.add_argument("--calib-heads", type=int, default=4)
ap.add_argument("--alpha", type=float, default=9.0)
ap.add_argument("--rho", type=float, default=0.85)
ap.add_argument("--sink", type=int, default=640)
ap.add_argument("--sink-gain", type=float, default=1.0)
ap.add_argument("--targets", default="0.21,0.42,0.57,0.77")
ap.add_argument("--iters", type=int, default=9)
args = ap.parse_args()

dev = torch.device("cuda", 0)
FRAMES, PER = 21, 3600
S, D = FRAMES * PER, 128


def make_base(H, seed=1234):
    g = torch.Generator(device=dev).manual_seed(seed)
    z = torch.randn(FRAMES, H, D, device=dev, generator=g)
    u = torch.empty_like(z)
    u[0] = z[0]
    for f in range(1, FRAMES):
        u[f] = args.rho * u[f - 1] + (1 - args.rho ** 2) ** 0.5 * z[f]
    u = u / u.norm(dim=-1, keepdim=True)
    fidx = torch.arange(S, device=dev) // PER
    cen = u[fidx]                                                    # (S,H,D)
    q0 = args.alpha * cen + torch.randn(S, H, D, device=dev, generator=g)
    k0 = args.alpha * cen + torch.randn(S, H, D, device=dev, generator=g)
    # global anchor keys at the END of the sequence (walked first): aligned with every frame's centroid mean
    anchor = u.mean(0)
    anchor = anchor / anchor.norm(dim=-1, keepdim=True)
    q0 = q0 + args.alpha * args.sink_gain * anchor
    k0[S - args.sink:] = k0[S - args.sink:] + args.alpha * (1 + args.sink_gain) * anchor
    v0 = torch.randn(S, H, D, device=dev, generator=g)
    return q0[None], k0[None], v0[None]


def qkv_at_factory(base):
    def qkv_at(t):
        s = 0.5 + (0.05 - 0.5) * t / max(1, args.steps - 1)
        g = torch.Generator(device=dev).manual_seed(10 ** 6 + t)
        out = []
        for x in base:
            n = torch.randn(x.shape, device=dev, generator=g)
            out.append(((1 - s * s) ** 0.5 * x + s * n).to(torch.bfloat16))
        return out
    return qkv_at


def schedule_efficiency(lists, H):
    """Greedy in-order list scheduling of the per-row tile counts onto 8 XCDs x 64 slots with the kernel's
    block->work map; returns sum(work) / (512 * makespan): 1.0 = perfectly balanced."""
    import heapq
    body = lists[:1].to(torch.int64)
    pairs = body[..., 1:1 + 2 * ((body.shape[-1] - 1) // 2)].unflatten(-1, (-1, 2))
    sizes = (pairs[..., 0] - pairs[..., 1] + 1).clamp_min(0)
    nr = (body[..., 0].clamp_min(2) // 2)
    live = torch.arange(pairs.shape[-2], device=body.device) < nr.unsqueeze(-1)
    counts = (sizes * live).sum(-1).flatten().cpu().tolist()         # order: (b, h, m) = vid
    n = len(counts); C = 64; full = (n // (8 * C)) * (8 * C)
    per_xcd = [[] for _ in range(8)]
    for bid in range(n):
        if bid >= full: vid = bid
        else:
            xcd, idx = bid & 7, bid >> 3
            vid = ((idx // C) * 8 + xcd) * C + (idx % C)
        per_xcd[bid & 7].append(counts[vid] + 6)                       # +6 tiles ~ prologue/epilogue cost
    makespan = 0
    for x in range(8):
        heap = [0] * 64
        for w in per_xcd[x]:
            t = heapq.heappop(heap); heapq.heappush(heap, t + w)
        makespan = max(makespan, max(heap))
    return sum(sum(v) for v in per_xcd) / (512.0 * makespan)


def run(thr, qkv_at, H, timed=False, errors=None):
    """errors: a list -> at a few steps the sparse output is compared with the dense kernel's on the same q, k, v
    (SURVEY.md 8d accuracy reporting: no reference tolerance exists for sparse outputs; the error grows with thr)."""
    att = L.LiteAttention(max_batch_size=1)
    att.threshold = thr
    trace, ms = [], []
    check = {0, 1, args.steps // 4, args.steps // 2, args.steps - 1} if errors is not None else set()
    for t in range(args.steps):
        q, k, v = qkv_at(t)
        trace.append(att.get_skip_fraction(batch=1) if att._skip_list is not None else 0.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = att(q, k, v); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
        if t in check:
            ref = L.flash_attn_func(q, k, v).float()
            d = (out.float() - ref).abs()
            errors.append({"step": t, "read_list_sparsity": round(trace[-1], 4), "max_abs": d.max().item(),
                           "mean_abs": d.mean().item(), "ref_max_abs": ref.abs().max().item(),
                           "ref_mean_abs": ref.abs().mean().item()})
            del ref, d
    global last_att
    last_att = att
    return trace, ms


last_att = None
targets = [float(x) for x in args.targets.split(",")]
res = {"config": vars(args), "S": S, "D": D, "tiles": list(L.get_tile_sizes(128, 2)), "device": torch.cuda.get_device_name(0)}
base_c = make_base(args.calib_heads)
qkv_c = qkv_at_factory(base_c)
found = {}
for tgt in targets:
    lo, hi, best = float(os.environ.get("LA_THR_LO", -60.0)), float(os.environ.get("LA_THR_HI", -1e-3)), None
    for _ in range(args.iters):
        mid = 0.5 * (lo + hi)
        trace, _ = run(mid, qkv_c, args.calib_heads)
        got = trace[-1]
        if best is None or abs(got - tgt) < abs(best[1] - tgt):
            best = (mid, got)
        if abs(got - tgt) <= 0.01:
            break
        if got < tgt: lo = mid
        else: hi = mid
    found[tgt] = best
    print(f"target {tgt:.2f}: thr={best[0]:.3f} -> step-{args.steps-1} read-list sparsity {best[1]:.3f} (calibration heads)", flush=True)
del base_c
base = make_base(args.heads)
qkv = qkv_at_factory(base)
dense_att = L.LiteAttention(enable_skipping=False)
dense_ms = []
for t in range(min(args.steps, 10)):
    q, k, v = qkv(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dense_att(q, k, v); e1.record(); torch.cuda.synchronize(); dense_ms.append(e0.elapsed_time(e1))
dense = sorted(dense_ms)[len(dense_ms) // 2]
res["dense_ms_per_step"] = dense
res["runs"] = []
for tgt in targets:
    thr = found[tgt][0]
    errs = []
    trace, ms = run(thr, qkv, args.heads, errors=errs)
    tot = sum(ms)
    eff = schedule_efficiency(last_att.current_read_list(), args.heads)
    res["runs"].append({"target": tgt, "thr": thr, "schedule_efficiency_last_list": eff, "sparsity_last_step": trace[-1], "mean_sparsity": sum(trace) / len(trace),
                        "sparsity_trace": [round(x, 4) for x in trace], "total_ms_50_steps": tot, "ms_last_step": ms[-1],
                        "speedup_vs_dense_total": dense * args.steps / tot, "t_last_over_dense": ms[-1] / dense,
                        "error_vs_dense_kernel": errs})
    print("   error vs dense: " + ", ".join(f"step {e['step']}: max {e['max_abs']:.3e} mean {e['mean_abs']:.3e}" for e in errs), flush=True)
    print(f"H={args.heads} target {tgt:.2f} thr {thr:.3f}: last-step sparsity {trace[-1]:.3f}, mean {sum(trace)/len(trace):.3f}, "
          f"{tot:.0f} ms / {args.steps} steps (dense {dense*args.steps:.0f} ms), last step {ms[-1]:.1f} ms vs dense {dense:.1f}, sched-eff {eff:.3f}", flush=True)
out = os.path.join(ROOT, "gpurun_out", f"{args.tag}_denoise50.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print("wrote", out)
