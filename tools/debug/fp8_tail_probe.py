"""GPU box: what the fp8 kernel's form of P costs in accuracy where the TAIL of the softmax carries mass: the step-49 tensors and READ lists
of the 50-step run (S = 75 600, H = 40, anchor keys + diffuse frames: tests/test_gpu_denoise_lists.py) and a dense randn case, sampled rows
against fp32 torch. One line per (library variant, LA_FP8_P setting): max |O - ref|, its tolerance, max |LSE - ref|, ms.
    LITEATTENTION_AMD_LIB=build_variants/f8_tau2.so python tools/debug/fp8_tail_probe.py [thr]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import liteattention_amd as L
from tools import selfcheck as sc
S, H, D = 75600, 40, 128
F8 = torch.float8_e4m3fn
thr = float(sys.argv[1]) if len(sys.argv) > 1 else -4.22
wl = sc.DenoiseWorkload(H, torch.device("cuda", 0))
att = L.LiteAttention(threshold=thr, max_batch_size=1)
for t in range(wl.steps - 1):
    q, k, v = wl.qkv(t); att(q, k, v); del q, k, v
q, k, v = wl.qkv(wl.steps - 1)
read = att.current_read_list().clone()
bm, bn = L.get_tile_sizes(D, 1)
kt = -(-S // bn)
q8, k8, v8 = [x.to(F8) for x in (q, k, v)]
must_do = torch.zeros(kt + 1, dtype=torch.int32, device="cuda"); must_do[0] = 2
name = os.path.basename(os.environ.get("LITEATTENTION_AMD_LIB", "tree"))
for mode in ("default", "exp", "rowsum"):
    os.environ.pop("LA_FP8_P", None)                      # "rowsum" = the default form (the reference's arithmetic)
    if mode == "exp": os.environ["LA_FP8_P"] = "mfma_rowsum"
    if mode not in ("exp", "rowsum"): os.environ["LA_FP8_P"] = "encoded"
    wr = torch.full_like(read, -7)
    f = lambda: L.flash_attn_func(q8, k8, v8, attn_read_list=read, attn_must_do_list=must_do, attn_write_list=wr, thr=float("-inf"), return_softmax_lse=True)
    out, lse = f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 4 * 1e3
    res = sc.sampled_row_check(q8, k8, v8, out, lse, read, bm, bn, heads=(0, 17, 39), n_rows=256, o_rtol=0.05, o_atol=1e-3, lse_atol=2e-2)
    print(f"{name:16s} {mode:8s} real lists thr {thr}: max|O-ref| {res['max_err']:.4f} (tol {res['tol']:.4f})  max|LSE-ref| {res['max_err_lse']:.4f}  {ms:.2f} ms", flush=True)
