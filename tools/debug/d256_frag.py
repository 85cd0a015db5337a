"""Debug: locate the d256 mismatch vs the oracle on fragmented lists (tests/test_gpu_fragmented.py, step 2)."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fragmented_qkv
from oracle import oracle as orc
import liteattention_amd as L
from liteattention_amd.flash_attn_interface import mha_fwd
from liteattention_amd import _cabi

D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4
B, S, thr, steps = 1, 16300, -3.0, 6
bm, bn = L.get_tile_sizes(D, 2)
Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
att = L.LiteAttention(threshold=thr, max_batch_size=B)
md_row = orc.expand_must_do_ref([0, 0], bn, Kt + 1)
for step in range(4):
    q, k, v = [x.bfloat16() for x in fragmented_qkv(B, S, H, D, seed=5, step=step, steps=steps, dtype=torch.float32)]
    rd_idx = att._phase if att._skip_list is not None else 0
    out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, must_do_list=md_row, thr=thr)
    d = (out.float().cpu() - o_ref).abs()
    idx = torch.nonzero(d == d.max())[0].tolist()
    b_, s_, h_, d_ = idx
    print(f"step {step}: max err {d.max().item():.5f} at row {s_} (q-tile {s_ // bm}, row in tile {s_ % bm}) head {h_} d {d_}: gpu {out[b_, s_, h_, d_].item():.6f} "
          f"ref {o_ref[b_, s_, h_, d_].item():.6f}; lse err {(lse.cpu() - lse_ref).abs().max().item():.2e}; n>tol {(d > 2**-8 * o_ref.abs().max() + 1e-3).sum().item()}")
    row_err = d[0, :, :, :].amax(-1)                      # [S, H]
    bad = torch.nonzero(row_err > 2**-8 * o_ref.abs().max() + 1e-3)
    print("   bad rows (row, head):", bad[:12].tolist(), "list row L:", [int(rd[0, h, r // bm, 0]) for r, h in bad[:12].tolist()])
    if len(bad):
        r, h = bad[0].tolist()
        dd = d[0, r, h]
        print("   err over d of first bad row: max", dd.max().item(), "n>1e-3:", (dd > 1e-3).sum().item(), "|o| at max", o_ref[0, r, h, dd.argmax()].item())
    # variants on the same read list
    for name, kw, flags in (("static", dict(_static_sched=True), 0), ("exact_rescale", {}, _cabi.LA_FLAG_EXACT_RESCALE), ("128row", {}, _cabi.LA_FLAG_KERNEL_128ROW)):
        try:
            if name == "static":
                o2, l2, *_ = mha_fwd(q.cuda(), k.cuda(), v.cuda(), attn_read_list=rd.cuda(), attn_must_do_list=md_row.cuda(),
                                     attn_write_list=torch.zeros_like(rd).cuda(), thr=thr, _must_do_is_1d=True, **kw)
            elif name == "exact_rescale":
                os.environ["LA_RESCALE_TAU"] = "0"
                o2, l2, *_ = mha_fwd(q.cuda(), k.cuda(), v.cuda(), attn_read_list=rd.cuda(), attn_must_do_list=md_row.cuda(),
                                     attn_write_list=torch.zeros_like(rd).cuda(), thr=thr, _must_do_is_1d=True)
                del os.environ["LA_RESCALE_TAU"]
            else:
                continue
            d2 = (o2.float().cpu() - o_ref).abs()
            print(f"   {name}: max err {d2.max().item():.5f}; equal to dynamic: {torch.equal(o2, out)}")
        except Exception as e:
            print("   ", name, "failed", repr(e)[:200])
