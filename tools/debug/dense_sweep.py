import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import liteattention_amd as L
for S, H in [(16384, 40), (16384, 80), (16384, 160), (32768, 40), (32768, 80), (75600, 40), (8192, 160)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g).bfloat16() for _ in range(3)]
    for _ in range(3): L.flash_attn_func(q, k, v)
    n = max(4, int(0.4 / (4 * H * S * S * 128 / 1.3e15)))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); L.flash_attn_func(q, k, v); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    med = ms[len(ms) // 2]
    items = H * -(-S // 256)
    print(f"S={S} H={H}: items {items} ({items / 256:.2f} rounds, {-(-S // 64)} tiles each): median {med:.3f} ms {4 * H * S * S * 128 / med / 1e9:.0f} TF (min {ms[0]:.3f}, n={n})")
    del q, k, v
