"""GPU box: how much sparsity does a given threshold find at different tile geometries? (this build votes over
256 query rows x 64 keys, the reference's Hopper kernel over 128 x 176, the hipcc-scheduled A/B kernel over 128 x 64.)

A torch restatement of the skip vote and the list evolution at an arbitrary (M, N) on the 50-step workload of BASELINE.json configs[2]
(tools.selfcheck.DenoiseWorkload, S = 75 600, a few heads): per step and head the full score matrix in q-tile slabs, per
(row, k-tile) maxima, the running maximum over the LISTED tiles in descending order, flag = AND over the M rows of
(m_loc - m_prev) c <= thr (softmax.h:190-194), next list = listed and (not flagged, or first flagged tile behind a kept one)
(contiguous listed tiles are treated as one range, tile Kt - 1 is never dropped). No outputs are computed.

    python tools/tile_geometry_sim.py [heads=4] [steps=50]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.selfcheck import DenoiseWorkload  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 4
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 50
GEOMS = [(128, 176, "reference Hopper tile"), (256, 64, "this build (x64 kernels)"), (128, 64, "LA_FLAG_KERNEL_128ROW")]
THRS = [-10.0, -6.0, -4.22, -3.0, -2.46]
dev = torch.device("cuda", 0)
wl = DenoiseWorkload(H, dev, steps=STEPS)
S, D = wl.S, wl.D
c = D ** -0.5 * 1.4426950408889634
listed = {}
for (M, N, _) in GEOMS:
    Qt, Kt = -(-S // M), -(-S // N)
    for thr in THRS:
        listed[(M, N, thr)] = torch.ones(H, Qt, Kt, dtype=torch.bool, device=dev)
hist = {k: [] for k in listed}
for t in range(STEPS):
    q, k, v = wl.qkv(t)
    for h in range(H):
        kh = k[0, :, h].float()
        for (M, N, _) in GEOMS:
            Qt, Kt = -(-S // M), -(-S // N)
            pad_k = Kt * N - S
            QB = 8 if M == 256 else 16                       # q-tiles per slab
            for q0 in range(0, Qt, QB):
                q1 = min(Qt, q0 + QB)
                rows = q[0, q0 * M: min(S, q1 * M), h].float()
                s = rows @ kh.T                                # [rows, S] fp32 scores (unscaled)
                if pad_k:
                    s = torch.nn.functional.pad(s, (0, pad_k), value=float("-inf"))
                tm = s.view(s.shape[0], Kt, N).amax(-1)       # per (row, k-tile) maximum
                pad_q = (q1 - q0) * M - tm.shape[0]
                if pad_q:                                      # rows past seqlen_q are zero rows in the kernel: score 0 against every key
                    tm = torch.cat([tm, torch.zeros(pad_q, Kt, device=dev)], 0)
                tm = tm.view(q1 - q0, M, Kt)
                for thr in THRS:
                    L = listed[(M, N, thr)][h, q0:q1]          # [qb, Kt]
                    masked = torch.where(L[:, None, :], tm, torch.full_like(tm, float("-inf")))
                    run = torch.cummax(masked.flip(-1), dim=-1).values.flip(-1)          # max over listed tiles >= this index
                    m_prev = torch.cat([run[..., 1:], torch.full_like(run[..., :1], float("-inf"))], -1)
                    vote_do = ((tm - m_prev) * c > thr)        # row says "do"
                    flag = ~vote_do.any(dim=1) & L             # every row at least |thr| bits below its running max
                    flag[:, Kt - 1] = False                    # the first walked tile is never flagged
                    kept_plain = L & ~flag
                    prev_kept = torch.cat([kept_plain[:, 1:], torch.zeros_like(kept_plain[:, :1])], -1)   # the tile walked just before (index + 1)
                    listed[(M, N, thr)][h, q0:q1] = L & (~flag | prev_kept)
            del s, tm
    for key, Lm in listed.items():
        hist[key].append(1.0 - Lm.float().mean().item())
    del q, k, v
print(f"sparsity of the list the NEXT step reads, after steps 10 / 25 / {STEPS} ({H} heads, S = {S}):")
print("| thr (log2) | " + " | ".join(f"{M} x {N} ({name})" for M, N, name in GEOMS) + " |")
print("|---|" + "---|" * len(GEOMS))
for thr in THRS:
    cells = []
    for (M, N, _) in GEOMS:
        hh = hist[(M, N, thr)]
        cells.append(" / ".join(f"{100 * hh[min(i, len(hh) - 1)]:.1f} %" for i in (9, 24, STEPS - 1)))
    print(f"| {thr} | " + " | ".join(cells) + " |")

# validation of the restatement: the kernel itself on the same 4-head workload (its tiles are 256 x 64)
import liteattention_amd as L  # noqa: E402
for thr in (-4.22, -2.46):
    att = L.LiteAttention(threshold=thr, max_batch_size=1)
    for t in range(STEPS):
        q, k, v = wl.qkv(t)
        att(q, k, v)
    print(f"kernel (256 x 64), thr {thr}: {100 * att.get_skip_fraction(batch=1):.1f} % after {STEPS} steps (restatement: {100 * hist[(256, 64, thr)][-1]:.1f} %)")
