"""Debug: the sampled-row check on the step-49 lists at thr -2.46 (tests/test_gpu_denoise_lists.py): where is the error, and how
large is the effect of rounding P to bf16 in a torch restatement?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import liteattention_amd as L
from liteattention_amd import selfcheck as sc
S, H, D = 75600, 40, 128
thr = float(sys.argv[1]) if len(sys.argv) > 1 else -2.462
wl = sc.DenoiseWorkload(H, torch.device("cuda", 0))
att = L.LiteAttention(threshold=thr, max_batch_size=1)
for t in range(wl.steps - 1):
    q, k, v = wl.qkv(t); att(q, k, v)
q, k, v = wl.qkv(wl.steps - 1)
read = att.current_read_list().clone()
out, lse = att(q, k, v, return_softmax_lse=True)
bm, bn = L.get_tile_sizes(D, 2)
rows = sc.sample_rows(S, 256, bm).cuda()
for h in (0, 17, 39):
    kf, vf = k[0, :, h].float(), v[0, :, h].float()
    s = (q[0, rows, h].float() @ kf.T) * D ** -0.5
    lists_h = read[0, h].cpu()
    for i, m in enumerate((rows // bm).tolist()):
        s[i].masked_fill_(~sc.listed_key_mask(lists_h[m].tolist(), bn, S, q.device), float("-inf"))
    p = torch.softmax(s, -1)
    ref = p @ vf
    mx = s.amax(-1, keepdim=True)
    pe = torch.exp(s - mx)
    ref_pt = (pe.bfloat16().float() @ vf) / pe.sum(-1, keepdim=True)
    err = (out[0, rows, h].float() - ref).abs()
    i, d_ = divmod(int(err.argmax()), D)
    print(f"head {h}: max|ref| {ref.abs().max().item():.4f} max err {err.max().item():.5f} at row {rows[i].item()} d {d_} (|ref| there {ref[i, d_].abs().item():.4f}, max p of the row {p[i].max().item():.4f}); "
          f"|ref_bf16P - ref| max {(ref_pt - ref).abs().max().item():.5f}; err vs ref_bf16P {(out[0, rows, h].float() - ref_pt).abs().max().item():.5f}; "
          f"tol now {2**-8 * ref.abs().max().item() + 1e-4:.5f}; lse err {(lse[0, h, rows] - torch.logsumexp(s, -1)).abs().max().item():.2e}")
