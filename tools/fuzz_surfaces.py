"""GPU box: randomized soak of the boundary's OTHER surfaces (companion of tools/fuzz_parity.py), mostly bit-exact self-consistency
relations that need no oracle:

  windows     LiteAttention.call_windowed over a random partition of the q-tiles == one __call__ (O, LSE, write list), bf16 / fp16 / fp8
  packed      lists + cu_seqlens in one launch == per-sequence fixed-length launches with the same lists (O, LSE, write lists)
  descales    fp8 with random per-(batch, K/V head) descales and GQA through LiteAttention vs the oracle with the same descales
  splits      SeqParallelLiteAttention over random K/V splits + flash_attn_combine vs one full call (fp32 merge, LSE)
  static      LA_SCHED=static == dynamic (O, LSE, write list) on lists produced by a few steps

    python tools/fuzz_surfaces.py [n_cases] [seed]"""
import math
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import structured_qkv, fragmented_qkv, fp8_lse_tol, fp8_p_round  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import liteattention_amd as L  # noqa: E402
from liteattention_amd.flash_attn_interface import mha_fwd  # noqa: E402

F8 = torch.float8_e4m3fn
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails, t_start = [], time.time()


def same_rows(a, b):
    n = int(b[..., 0].max())
    live = torch.arange(n + 1, device=b.device) <= b[..., 0:1]
    return bool(((a[..., : n + 1] == b[..., : n + 1]) | ~live).all())


def make(dtype, B, S, H, D, seed, step=0, gen="structured"):
    if gen == "structured":
        q, k, v = structured_qkv(B, S, H, D, seed=seed, alpha=8.0 - 0.5 * step, dtype=torch.float32)
    else:
        q, k, v = fragmented_qkv(B, S, H, D, seed=seed % 1000, step=step, steps=4, dtype=torch.float32)
    cast = {"fp8": lambda x: x.to(F8), "fp16": lambda x: x.half(), "bf16": lambda x: x.bfloat16()}[dtype]
    return cast(q), cast(k), cast(v)


for case in range(n_cases):
    kind = rng.choice(["windows", "packed", "descales", "splits", "static"])
    dtype = rng.choice(["bf16", "fp16", "fp8"])
    D = rng.choice([64, 96, 128, 192, 256]) if dtype == "fp8" else rng.choice([64, 96, 128, 192, 256])
    seed = rng.randrange(1 << 20)
    desc = f"case {case}: {kind} {dtype} D{D} seed {seed}"
    for v_ in ("LA_SCHED",):
        os.environ.pop(v_, None)
    try:
        es = 1 if dtype == "fp8" else 2
        bm, bn = L.get_tile_sizes(D, es)
        thr = rng.choice([-1.5, -2.5, -4.0])
        gen = rng.choice(["structured", "fragmented"])
        if kind == "windows":
            B, S, H = rng.choice([1, 2]), rng.choice([700, 1300, 2600, 4100]), rng.choice([1, 3])
            desc += f" B{B} S{S} H{H} thr {thr}"
            Qt = math.ceil(S / bm)
            a1, a2 = L.LiteAttention(threshold=thr, max_batch_size=B), L.LiteAttention(threshold=thr, max_batch_size=B)
            for step in range(3):
                q, k, v = [x.cuda() for x in make(dtype, B, S, H, D, seed, step, gen)]
                o1, l1 = a1(q, k, v, return_softmax_lse=True)
                from liteattention_amd.flash_attn_interface import q_tiles_per_item
                u = q_tiles_per_item(D, es)              # LA_VOTE=half: a workgroup item is two q-tiles; windows hold whole items
                cuts = sorted(set([0, Qt] + [rng.randrange(0, Qt + 1) // u * u for _ in range(rng.randrange(0, 4))]))
                wins = [(cuts[i], cuts[i + 1] - cuts[i]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]
                o2, l2 = a2.call_windowed(q, k, v, wins, return_softmax_lse=True)
                if not (torch.equal(o1, o2) and torch.equal(l1, l2) and same_rows(a2.current_read_list(), a1.current_read_list())):
                    fails.append(f"{desc} | step {step} windows {wins}: windowed != single launch")
                    break
        elif kind == "static":
            B, S, H = rng.choice([1, 2]), rng.choice([900, 2100, 5000]), rng.choice([2, 5])
            desc += f" B{B} S{S} H{H} thr {thr}"
            a1, a2 = L.LiteAttention(threshold=thr, max_batch_size=B), L.LiteAttention(threshold=thr, max_batch_size=B)
            for step in range(3):
                q, k, v = [x.cuda() for x in make(dtype, B, S, H, D, seed, step, gen)]
                os.environ.pop("LA_SCHED", None)
                o1, l1 = a1(q, k, v, return_softmax_lse=True)
                os.environ["LA_SCHED"] = "static"
                o2, l2 = a2(q, k, v, return_softmax_lse=True)
                if not (torch.equal(o1, o2) and torch.equal(l1, l2) and same_rows(a2.current_read_list(), a1.current_read_list())):
                    fails.append(f"{desc} | step {step}: static != dynamic")
                    break
        elif kind == "packed":
            if D > 128 and dtype != "fp8":          # (lists + cu_seqlens above head_dim 128: the fp8 kernels only)
                D = 128
                bm, bn = L.get_tile_sizes(D, es)
            nseq, H = rng.choice([2, 3, 5]), rng.choice([1, 2])
            lens_q = [rng.choice([0, 1, 40, 256, 300, 700, 1200]) for _ in range(nseq)]
            lens_k = [rng.choice([0, 1, 13, 64, 130, 500, 900, 1500]) for _ in range(nseq)]
            if max(lens_q) == 0:
                lens_q[0] = 300
            if max(lens_k) == 0:
                lens_k[0] = 200
            desc += f" lens_q {lens_q} lens_k {lens_k} H{H} thr {thr}"
            qs, ks, vs = [], [], []
            for b in range(nseq):
                q, k, v = make(dtype, 1, max(lens_q[b], lens_k[b], 1), H, D, seed + b, 0, gen)
                qs.append(q[0, : lens_q[b]]); ks.append(k[0, : lens_k[b]]); vs.append(v[0, : lens_k[b]])
            qp, kp, vp = torch.cat(qs).cuda(), torch.cat(ks).cuda(), torch.cat(vs).cuda()
            cq = [0] + torch.tensor(lens_q).cumsum(0).tolist()
            ck = [0] + torch.tensor(lens_k).cumsum(0).tolist()
            cq_d, ck_d = torch.tensor(cq, dtype=torch.int32).cuda(), torch.tensor(ck, dtype=torch.int32).cuda()
            Qt, Kt = math.ceil(max(lens_q) / bm), math.ceil(max(lens_k) / bn)
            lists = torch.zeros(2, nseq, H, Qt, Kt + 1, dtype=torch.int32)
            for b in range(nseq):
                lists[:, b, :, :, 0] = 2
                lists[:, b, :, :, 1] = max(math.ceil(lens_k[b] / bn) - 1, 0)
            lists = lists.cuda()
            must_do = torch.tensor([2, 0, 0], dtype=torch.int32).cuda()
            rd = 0
            for step in range(2):
                lists[1 - rd].fill_(-7)
                o, lse, *_ = mha_fwd(qp, kp, vp, cu_seqlens_q=cq_d, cu_seqlens_k=ck_d, max_seqlen_q=max(lens_q), max_seqlen_k=max(lens_k),
                                     attn_read_list=lists[rd], attn_must_do_list=must_do, attn_write_list=lists[1 - rd], thr=thr, _must_do_is_1d=True)
                ok = bool(torch.isfinite(o.float()).all())
                for b in range(nseq):
                    if lens_q[b] == 0:
                        continue
                    sq = slice(cq[b], cq[b + 1])
                    qt_b, kt_b = math.ceil(lens_q[b] / bm), math.ceil(lens_k[b] / bn)
                    kw = {}
                    if lens_k[b] > 0:
                        rd_b = lists[rd, b: b + 1, :, :qt_b, : kt_b + 1].contiguous()
                        wr_b = torch.zeros_like(rd_b)
                        kw = dict(attn_read_list=rd_b, attn_must_do_list=must_do, attn_write_list=wr_b, thr=thr, _must_do_is_1d=True)
                    o_b, lse_b, *_ = mha_fwd(qs[b][None].cuda(), ks[b][None].cuda(), vs[b][None].cuda(), **kw)
                    ok = ok and torch.equal(o[sq], o_b[0]) and torch.equal(lse[:, sq], lse_b[0])
                    if lens_k[b] > 0 and kt_b > 1:
                        ok = ok and same_rows(lists[1 - rd, b, :, :qt_b, : kt_b + 1], wr_b[0])
                if not ok:
                    fails.append(f"{desc} | step {step}: packed launch != per-sequence launches")
                    break
                rd = 1 - rd
        elif kind == "descales":
            dtype, D = "fp8", rng.choice([64, 96, 128, 192, 256])
            bm, bn = L.get_tile_sizes(D, 1)
            B, Hk = rng.choice([1, 2, 3]), rng.choice([1, 2])
            H = Hk * rng.choice([1, 2, 4])
            Sq, Sk = rng.choice([100, 300, 1000]), rng.choice([64, 333, 1400, 2500])
            desc = f"case {case}: descales fp8 D{D} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} thr {thr} seed {seed}"
            g = torch.Generator().manual_seed(seed)
            qd, kd, vd = [(0.3 + 1.5 * torch.rand(B, Hk, generator=g)) for _ in range(3)]
            att = L.LiteAttention(threshold=thr, max_batch_size=B)
            Qt, Kt = math.ceil(Sq / bm), math.ceil(Sk / bn)
            md_row = orc.expand_must_do_ref([0, 0], bn, max(Kt + 1, 3))
            margins = torch.empty(B, H, Qt, Kt)
            for step in range(2):
                q, k, v = make("fp8", B, max(Sq, Sk), H, D, seed, step, gen)
                q, k, v = q[:, :Sq], k[:, :Sk, :Hk], v[:, :Sk, :Hk]
                rd_idx = att._phase if att._skip_list is not None else 0
                out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, q_descale=qd.cuda(), k_descale=kd.cuda(), v_descale=vd.cuda())
                rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
                wr_orc = torch.zeros_like(wr)
                o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, must_do_list=md_row, thr=thr,
                                                   margins=margins, p_round=fp8_p_round(), q_descale=qd, k_descale=kd, v_descale=vd)
                eo = (out.float().cpu() - o_ref).abs().max().item()
                el = (lse.cpu() - lse_ref).abs().max().item()
                from test_gpu_parity import _compare_lists
                bad, border = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
                if not (eo <= 0.05 * o_ref.abs().max().item() + 2e-2 and el <= fp8_lse_tol() and bad == 0 and border <= 3):
                    fails.append(f"{desc} | step {step}: O err {eo:.4g} LSE err {el:.4g} list rows bad {bad} borderline {border}")
                    break
        else:   # splits
            if dtype == "fp8":
                dtype = "bf16"
            B, S, H = 1, rng.choice([1024, 2048, 3072]), rng.choice([1, 2])
            nsp = rng.choice([2, 3, 4])
            desc += f" S{S} H{H} splits {nsp}"
            q, k, v = [x.cuda() for x in make(dtype, B, S, H, D, seed, 0, gen)]
            bounds = sorted(set([0, S] + [rng.randrange(1, S) for _ in range(nsp - 1)]))
            sp = L.SeqParallelLiteAttention(num_nodes=len(bounds) - 1, threshold=-30.0, max_batch_size=B)
            outs, lses = [], []
            for j in range(len(bounds) - 1):
                o, l = sp(q, k[:, bounds[j]: bounds[j + 1]], v[:, bounds[j]: bounds[j + 1]], split_idx=j, return_softmax_lse=True)
                outs.append(o.float()); lses.append(l)
            om, lm = L.flash_attn_combine(torch.stack(outs), torch.stack(lses))
            of, lf = L.flash_attn_func(q, k, v, return_softmax_lse=True)
            eo = (om - of.float()).abs().max().item()
            el = (lm - lf).abs().max().item()
            ulp = 2.0 ** -10 if dtype == "fp16" else 2.0 ** -7
            if not (eo <= 2 * ulp * of.float().abs().max().item() + 1e-3 and el <= 1e-3):
                fails.append(f"{desc} bounds {bounds}: merged splits vs full call O err {eo:.4g} LSE err {el:.4g}")
    except Exception as e:  # noqa: BLE001
        fails.append(f"{desc} | EXCEPTION {e!r}")
    if case % 10 == 9:
        print(f"... {case + 1} cases, {len(fails)} failures, {time.time() - t_start:.0f} s", flush=True)
for f_ in fails:
    print("FAIL", f_)
print(f"fuzz_surfaces: {n_cases} cases, {len(fails)} failures")
sys.exit(1 if fails else 0)
