"""GPU box, 1 GPU: what a collective's kernels cost the windowed attention. No second GPU is
available to the builder, so the all-gather of `HeadShardedLiteAttention` is EMULATED by what it is on the device: copy kernels on
another stream that take CUs and memory bandwidth away while the next q-tile window computes. One rank's share of the headline
workload at G = 8 / 4 / 2 (H = 40 / G heads, S = 75 600, 42 % banded lists); per window count n: the step alone, the step with a
copy stream running the whole time (a continuous 256 MiB device-to-device copy loop: far MORE than a rank's all-gather moves), and
the step with exactly the per-window copies the overlapped form issues (window i's rows x (G - 1) peers, behind window i, on a side
stream) - for the dynamic (persistent workgroups) and the static-after-first scheduling the driver uses. Results are checked
bit-identical to the undisturbed run.  ->  profiles/r05_window_interference.md"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from liteattention_amd.parallel import plan_q_windows
from bench import banded_rows, impose_lists

S, D = 75600, 128
bm, bn = L.get_tile_sizes(D, 2)
Qt, Kt = -(-S // bm), -(-S // bn)
side = torch.cuda.Stream()
hog_src = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
hog_dst = torch.empty_like(hog_src)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


print("| G (heads per rank) | windows | scheduling | alone ms | under a continuous copy stream ms (slowdown) | with the per-window gather copies ms (slowdown) | results |")
print("|---|---|---|---|---|---|---|")
for G in (8, 4, 2):
    Hl = 40 // G
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, Hl, D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    att = L.LiteAttention(max_batch_size=1); att.threshold = float("-inf")
    att(q, k, v)
    impose_lists(att, banded_rows(Qt, Kt, bm, bn, 0.42))
    ref = att(q, k, v).clone()
    peers = torch.empty((G - 1, 1, S, Hl, D), dtype=torch.bfloat16, device="cuda")       # where the emulated gather lands
    for n in (1, 3, 6):
        w = plan_q_windows(Qt, Hl, n)
        for sched in ((False, "dynamic"),) if n == 1 else ((False, "dynamic"), ("after_first", "static after the first window")):
            run = lambda hook=None: att.call_windowed(q, k, v, w, hook, static_sched=sched[0])
            alone = timed(run)
            # (a) a copy stream that never stops
            def with_hog():
                with torch.cuda.stream(side):
                    for _ in range(24):                       # ~24 x 256 MiB queued beside one step
                        hog_dst.copy_(hog_src, non_blocking=True)
                return run()
            hog = timed(with_hog, reps=3)
            torch.cuda.synchronize()
            # (b) the copies the overlapped all-gather issues: window i's rows to G - 1 peers, behind window i, on the side stream
            def hook(i, out, r0, r1):
                e = torch.cuda.Event(); e.record()
                with torch.cuda.stream(side):
                    side.wait_event(e)
                    for p in range(G - 1):
                        peers[p, :, r0:r1].copy_(out[:, r0:r1], non_blocking=True)
            gat = timed(lambda: run(hook), reps=4)
            torch.cuda.synchronize()
            out = run(hook); torch.cuda.synchronize()
            ok = bool(torch.equal(out, ref)) and bool(torch.equal(peers[0], ref))
            print(f"| {G} ({Hl}) | {len(w)} | {sched[1]} | {alone:.2f} | {hog:.2f} ({hog / alone:.3f}x) | {gat:.2f} ({gat / alone:.3f}x) | {'bit-identical' if ok else 'DIFFER'} |")
    del q, k, v, peers
