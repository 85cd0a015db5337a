"""GPU box: randomized parity soak of the whole boundary against the CPU oracle (not a committed test: minutes of oracle time).

Each case draws (dtype, head_dim, B, Sq, Sk, H, Hk, threshold, steps, must-do list, softmax scale, work distribution) at random, runs
`steps` calls of LiteAttention / flash_attn_func on structured inputs (so that thresholds produce real lists), and compares every step
with the oracle on the SAME read list: O, LSE, write lists (bit-exact under the 1e-3 margin rule). Also: packed (cu_seqlens) form of
the same sequences == the per-sequence results. Prints one line per failure and a summary; exit code 1 on any failure.

    python tools/fuzz_parity.py [n_cases] [seed]"""
import math
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import structured_qkv, fragmented_qkv, fp8_lse_tol, fp8_lse_tol_vs_exact, fp8_p_round  # noqa: E402
from test_gpu_parity import _compare_lists  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import liteattention_amd as L  # noqa: E402

F8 = torch.float8_e4m3fn
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = []
t_start = time.time()
for case in range(n_cases):
    dtype = rng.choice(["bf16", "bf16", "fp16", "fp8", "fp8"])
    D = rng.choice([128, 128, 64, 48, 96, 80, 192, 256, 160]) if dtype == "fp8" else rng.choice([64, 96, 128, 128, 192, 256, 80, 160])      # e4m3: 64, 96, 128, 192, 256 natively (48 -> 64, 80 -> 96, 160 -> 192 zero-padded)
    B = rng.choice([1, 1, 2, 3])
    Hk = rng.choice([1, 2, 3])
    H = Hk * rng.choice([1, 1, 2, 4])
    Sq = rng.choice([1, 17, 255, 256, 257, 300, 777, 1024, 1500, 2300])
    Sk = rng.choice([1, 13, 64, 65, 200, 640, 1000, 1537, 2048, 3100])
    thr = rng.choice([-1.0, -2.0, -3.0, -6.0, float("-inf")])
    steps = rng.choice([1, 2, 3, 4])
    scale = rng.choice([None, None, 0.05, 0.15])
    static = rng.random() < 0.3
    md = rng.choice([None, None, "one", "two"])
    gen = rng.choice(["structured", "fragmented", "randn"])
    seed = rng.randrange(1 << 20)
    desc = f"case {case}: {dtype} D{D} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} thr {thr} steps {steps} scale {scale} static {static} md {md} {gen} seed {seed}"
    if os.environ.get("FUZZ_VERBOSE"):
        print("START", desc, flush=True)
    os.environ.pop("LA_SCHED", None)
    if static:
        os.environ["LA_SCHED"] = "static"
    try:
        es = 1 if dtype == "fp8" else 2
        bm, bn = L.get_tile_sizes(D, es)
        Qt, Kt = math.ceil(Sq / bm), math.ceil(Sk / bn)
        if dtype == "fp8":
            cast, p_round = (lambda x: x.to(F8)), fp8_p_round()
            tol = lambda o: 0.05 * o.abs().max().item() + 2e-2                        # noqa: E731
            lse_tol = fp8_lse_tol()
        elif dtype == "fp16":
            cast, p_round = (lambda x: x.half()), "f16"
            tol = lambda o: 2.0 ** -9 * o.abs().max().item() + 1e-3                   # noqa: E731
            lse_tol = 1e-3
        else:
            cast, p_round = (lambda x: x.bfloat16()), True
            tol = lambda o: 2.0 ** -7 * o.abs().max().item() + 1e-3                   # noqa: E731
            lse_tol = 1e-3
        toks = None
        if md == "one" and Sk > 130:
            toks = [Sk - 60, Sk // 2]
        elif md == "two" and Sk > 400:
            toks = [Sk - 10, Sk - 200, Sk // 3, 5]
        md_row = orc.expand_must_do_ref(toks if toks else [0, 0], bn, max(Kt + 1, 3))
        att = L.LiteAttention(threshold=thr if thr != float("-inf") else -1.0, max_batch_size=B)
        att.threshold = thr
        margins = torch.empty(B, H, Qt, Kt)
        for step in range(steps):
            S_ = max(Sq, Sk)
            if gen == "structured":
                q, k, v = structured_qkv(B, S_, H, D, seed=seed, alpha=8.0 - step, dtype=torch.float32)
            elif gen == "fragmented":
                q, k, v = fragmented_qkv(B, S_, H, D, seed=seed % 1000, step=step, steps=max(steps, 2), dtype=torch.float32)
            else:
                g = torch.Generator().manual_seed(seed + step)
                q, k, v = [torch.randn(B, S_, H, D, generator=g) for _ in range(3)]
            q, k, v = cast(q[:, :Sq]), cast(k[:, :Sk, :Hk]), cast(v[:, :Sk, :Hk])
            rd_idx = att._phase if att._skip_list is not None else 0
            out, lse = att(q.cuda(), k.cuda(), v.cuda(), scale=scale, return_softmax_lse=True, must_do_list=toks)
            rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
            wr_orc = torch.zeros_like(wr)
            o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, must_do_list=md_row,
                                                     thr=thr, margins=margins, p_round=p_round, softmax_scale=scale)
            eo = (out.float().cpu() - o_ref).abs().max().item()
            el = (lse.cpu() - lse_ref).abs()
            el = el[torch.isfinite(el)].max().item() if torch.isfinite(el).any() else 0.0
            bad, border = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
            # (round 4 carried a "byte-flip rule" here for the fp8 default form: the oracle encoded P~ relative to the true running maximum, the
            # kernel relative to its lazy reference maximum - two different 8-bit grids, so on peaked rows they disagreed by up to one byte
            # on a dominant key. Round 5: the oracle restates the kernel's grid exactly (p_round 4, lin_lazy; identical bytes on exact scores:
            # tests/test_gpu_fp8.py), and the rule is gone.)
            if dtype == "fp8" and fp8_p_round() == "fp8_lin":
                # the second, independent bound of the default fp8 form: against the EXACT LSE (un-rounded P) the encoding's own bound
                _, lse_x, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=torch.zeros_like(wr), must_do_list=md_row,
                                             thr=thr, p_round=False, softmax_scale=scale)
                ex = (lse.cpu() - lse_x).abs()
                ex = ex[torch.isfinite(ex)].max().item() if torch.isfinite(ex).any() else 0.0
                if ex > fp8_lse_tol_vs_exact():
                    el = max(el, 10.0 * lse_tol)               # fail the case
            if not (eo <= tol(o_ref) and el <= lse_tol and bad == 0 and border <= 3 and bool(torch.isfinite(out.float()).all())):
                fails.append(f"{desc} | step {step}: O err {eo:.4g} (tol {tol(o_ref):.4g}) LSE err {el:.4g} (tol {lse_tol:.3g}) list rows bad {bad} borderline {border}")
                break
        # packed form of the same batch (dense): every sequence of the batch as its own length
        if D in (64, 96, 128, 192, 256) and Sq > 1:
            lens_q = [max(1, Sq - 37 * b) for b in range(B)]
            lens_k = [max(1, Sk - 11 * b) for b in range(B)]
            qp = torch.cat([q[b, : lens_q[b]] for b in range(B)]).cuda()
            kp = torch.cat([k[b, : lens_k[b]] for b in range(B)]).cuda()
            vp = torch.cat([v[b, : lens_k[b]] for b in range(B)]).cuda()
            cq = [0] + torch.tensor(lens_q).cumsum(0).tolist()
            ck = [0] + torch.tensor(lens_k).cumsum(0).tolist()
            o_p = L.flash_attn_varlen_func(qp, kp, vp, cq, ck, softmax_scale=scale)
            for b in range(B):
                o_b = L.flash_attn_func(q[b: b + 1, : lens_q[b]].cuda(), k[b: b + 1, : lens_k[b]].cuda(), v[b: b + 1, : lens_k[b]].cuda(), softmax_scale=scale)
                if not torch.equal(o_p[cq[b]: cq[b + 1]], o_b[0]):
                    fails.append(f"{desc} | packed form differs from the fixed-length call at sequence {b}")
                    break
        # host split-KV of the dense call (round 6): forced 2 .. 4 splits against the unsplit launch, one more 16-bit rounding (the merge's)
        if Sk > 128 and dtype != "fp8":
            ns = rng.choice([2, 3, 4])
            o_1 = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), softmax_scale=scale)
            o_s = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), softmax_scale=scale, num_splits=ns)
            es_ = (o_s.float() - o_1.float()).abs().max().item()
            # Both results sit on the 16-bit grid. Unsplit: one rounding of the fp32 result (<= 1/2 ulp). Split: the partials are rounded (<= 1/2 ulp of
            # values no larger than the result's convex hull), merged in fp32 and rounded again: <= 1 ulp. The two can therefore land TWO grid steps apart
            # (1.5 ulp between them before the last rounding); one step is the common case. Bound: 2 ulp at the largest magnitude = 2^-6 of it for bf16
            # (2^-9 fp16) - and each result is held to its own bound against an fp32 torch attention.
            ulp = (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10) * o_1.float().abs().max().item()
            ref_ = torch.nn.functional.scaled_dot_product_attention(q.cuda().float().transpose(1, 2), k.cuda().float().repeat_interleave(H // Hk, 2).transpose(1, 2),
                                                                    v.cuda().float().repeat_interleave(H // Hk, 2).transpose(1, 2), scale=scale).transpose(1, 2)
            e1_, e2_ = (o_1.float() - ref_).abs().max().item(), (o_s.float() - ref_).abs().max().item()
            if not (es_ <= 2 * ulp + 1e-3 and e1_ <= ulp + 1e-3 and e2_ <= 1.5 * ulp + 1e-3):
                fails.append(f"{desc} | split-KV ({ns}) differs from the unsplit launch by {es_:.4g} (ulp at max|O| {ulp:.4g}; vs fp32: unsplit {e1_:.4g}, split {e2_:.4g})")
    except Exception as e:  # noqa: BLE001
        fails.append(f"{desc} | EXCEPTION {e!r}")
    if case % 10 == 9:
        print(f"... {case + 1} cases, {len(fails)} failures, {time.time() - t_start:.0f} s", flush=True)
for f_ in fails:
    print("FAIL", f_)
print(f"fuzz_parity: {n_cases} cases, {len(fails)} failures")
sys.exit(1 if fails else 0)
