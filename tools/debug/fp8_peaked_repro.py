import math, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import fragmented_qkv, fp8_lse_tol, fp8_p_round
from oracle import oracle as orc
import liteattention_amd as L
F8 = torch.float8_e4m3fn
B, Sq, Sk, H, Hk, D, thr, steps, scale, seed = 2, 1, 1537, 12, 3, 128, -3.0, 4, 0.15, 86089
bm, bn = L.get_tile_sizes(D, 1)
Qt, Kt = math.ceil(Sq / bm), math.ceil(Sk / bn)
att = L.LiteAttention(threshold=thr, max_batch_size=B)
md_row = orc.expand_must_do_ref([0, 0], bn, max(Kt + 1, 3))
margins = torch.empty(B, H, Qt, Kt)
for step in range(steps):
    q, k, v = fragmented_qkv(B, max(Sq, Sk), H, D, seed=seed % 1000, step=step, steps=max(steps, 2), dtype=torch.float32)
    q, k, v = q[:, :Sq].to(F8), k[:, :Sk, :Hk].to(F8), v[:, :Sk, :Hk].to(F8)
    rd_idx = att._phase if att._skip_list is not None else 0
    out, lse = att(q.cuda(), k.cuda(), v.cuda(), scale=scale, return_softmax_lse=True)
    rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
    wr_o = torch.zeros_like(wr)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_o, must_do_list=md_row, thr=thr, margins=margins, p_round=fp8_p_round(), softmax_scale=scale)
    o_exact, _, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=torch.zeros_like(wr), must_do_list=md_row, thr=thr, p_round=False, softmax_scale=scale)
    eo = (out.float().cpu() - o_ref).abs().max().item()
    ex = (out.float().cpu() - o_exact).abs().max().item()
    print(f"form {os.environ.get('LA_FP8_P','') or 'reference (default)'} step {step}: |O - oracle(same form)| {eo:.4f}  |O - fp32-P oracle| {ex:.4f}  tol {0.05 * o_ref.abs().max().item() + 2e-2:.4f}  max|O| {o_ref.abs().max().item():.3f}")
