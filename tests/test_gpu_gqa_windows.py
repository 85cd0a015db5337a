"""GPU: GQA / MQA (nheads_k < nheads) and q-tile windows (C-ABI 3) on every forward kernel.

Tolerances are those stated in test_gpu_parity.py / test_gpu_fp8.py (reference rule vs `out_ref`; 2^-8 max|O| + 1e-3 vs the
tiled oracle; LSE 1e-3; lists bit-exact up to borderline tiles)."""
import math
import os

import pytest
import torch

from helpers import GQA_CASES, GQA_FP8_CASES, load_dense_case, ref_tolerance, structured_qkv, fp8_lse_tol, fp8_p_round
from test_gpu_parity import _compare_lists, _oracle_tol

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def _L():
    import liteattention_amd as L
    return L


def _orc():
    from oracle import oracle as orc
    return orc


@pytest.mark.parametrize("name", GQA_CASES)
def test_gqa_dense_matches_reference_outputs(name):
    """bf16 d128 (x64 kernel) and d64 (v2 kernel) with nheads_k < nheads against the reference's attention_ref outputs."""
    L, orc = _L(), _orc()
    c = load_dense_case(name)
    D = c["D"]
    bm, bn = L.get_tile_sizes(D, 2)
    q, k, v = [x.to(torch.bfloat16).cuda() for x in (c["q"], c["k"], c["v"])]
    out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    assert out.shape == q.shape and lse.shape == (q.shape[0], q.shape[2], q.shape[1])
    err = (out.float().cpu() - c["out_ref"]).abs().max().item()
    assert err <= ref_tolerance(c["out_ref"], c["pt_maxerr"]), (err, ref_tolerance(c["out_ref"], c["pt_maxerr"]))
    assert (lse.cpu() - c["lse_ref"]).abs().max().item() <= 1e-3
    o_t, lse_t, _ = orc.qkskip_fwd(c["q"], c["k"], c["v"], block_m=bm, block_n=bn)
    assert (out.float().cpu() - o_t).abs().max().item() <= _oracle_tol(o_t)


@pytest.mark.parametrize("name", GQA_FP8_CASES)
def test_gqa_fp8_matches_reference_outputs(name):
    """fp8 with nheads_k < nheads: descales are (batch, nheads_k) for q, k and v (flash_api.cpp:689-691); the V^T
    workspace holds nheads_k heads."""
    L, orc = _L(), _orc()
    c = load_dense_case(name)
    q, k, v = [x.to(F8).cuda() for x in (c["q"], c["k"], c["v"])]
    qd, kd, vd = [c[n].cuda() for n in ("q_descale", "k_descale", "v_descale")]
    out, lse = L.flash_attn_func(q, k, v, q_descale=qd, k_descale=kd, v_descale=vd, return_softmax_lse=True)
    err = (out.float().cpu() - c["out_ref"]).abs().max().item()
    assert err <= ref_tolerance(c["out_ref"], c["pt_maxerr"]), (err, ref_tolerance(c["out_ref"], c["pt_maxerr"]))
    assert (lse.cpu() - c["lse_ref"]).abs().max().item() <= fp8_lse_tol()
    bm8, bn8 = L.get_tile_sizes(128, 1)
    o8, lse8, _ = orc.qkskip_fwd(c["q"], c["k"], c["v"], block_m=bm8, block_n=bn8, p_round=fp8_p_round(),
                                 q_descale=c["q_descale"], k_descale=c["k_descale"], v_descale=c["v_descale"])
    assert (out.float().cpu() - o8).abs().max().item() <= 0.05 * o8.abs().max().item() + 2e-2
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()


@pytest.mark.parametrize("D", [128, 64])
def test_gqa_skip_lists_match_oracle_over_steps(D):
    """QK-Skip with GQA: lists are per QUERY head ([B, H, Qt, Kt+1]); 4 steps, same read list fed to the oracle."""
    L, orc = _L(), _orc()
    B, S, H, Hk, thr = 1, 1536, 4, 2, -3.0
    bm, bn = L.get_tile_sizes(D, 2)
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    margins = torch.empty(B, H, Qt, Kt)
    listed = []
    for step in range(4):
        qk, k, v = structured_qkv(B, S, Hk, D, seed=31)
        # query heads 2j, 2j+1 read K/V head j: give them that head's frame centroids so attention stays structured
        q = qk.repeat_interleave(H // Hk, dim=2)
        g = torch.Generator().manual_seed(500 + step)
        q = (q.float() + 0.3 * torch.randn(q.shape, generator=g)).bfloat16()
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        assert tuple(rd.shape) == (B, H, Qt, Kt + 1)
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, thr=thr,
                                                 margins=margins)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0
        listed.append(orc.listed_tiles(wr[:B]))
    assert listed[-1] < 0.9 * B * H * Qt * Kt


# ------------------------------------------------------------------------------------------ q-tile windows
@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("bf16", 64), ("fp8", 128)])
def test_q_tile_windows_equal_one_launch_bit_exactly(dtype, D):
    """`LiteAttention.call_windowed` (one launch per q-tile window, C-ABI q_tile_begin/q_tile_count) must reproduce the
    single launch bit for bit: output, LSE and the written list, over several steps; the hook sees each window in order."""
    L = _L()
    B, S, H = 2, 1700, 3                      # ragged last q-tile
    es = 1 if dtype == "fp8" else 2
    bm, bn = L.get_tile_sizes(D, es)
    Qt = math.ceil(S / bm)
    q, k, v = structured_qkv(B, S, H, D, seed=9)
    if dtype == "fp8":
        q, k, v = [x.float().to(F8) for x in (q, k, v)]
    q, k, v = q.cuda(), k.cuda(), v.cuda()
    from liteattention_amd.flash_attn_interface import q_tiles_per_item
    u = q_tiles_per_item(D, es)                                  # LA_FLAG_HALF_VOTE (LA_VOTE=half): windows are pairs of 128-row q-tiles
    cuts = sorted({0, u, (Qt // 2) // u * u, Qt})
    windows = [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    assert len(windows) >= 2
    one = L.LiteAttention(threshold=-2.0, max_batch_size=B)
    win = L.LiteAttention(threshold=-2.0, max_batch_size=B)
    sta = L.LiteAttention(threshold=-2.0, max_batch_size=B)      # the multi-GPU driver's mix: window 0 tickets, later static
    for step in range(3):
        seen = []
        o1, l1 = one.call_windowed(q, k, v, [(0, Qt)], return_softmax_lse=True)
        o2, l2 = win.call_windowed(q, k, v, windows, lambda i, out, r0, r1: seen.append((i, r0, r1)),
                                   return_softmax_lse=True)
        o0, l0 = (L.LiteAttention(threshold=-2.0, max_batch_size=B)(q, k, v, return_softmax_lse=True) if step == 0
                  else (o1, l1))
        assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(o0, o1) and torch.equal(l0, l1)
        assert torch.equal(one._skip_list, win._skip_list)
        o3, l3 = sta.call_windowed(q, k, v, windows, return_softmax_lse=True, static_sched="after_first")
        assert torch.equal(o3, o1) and torch.equal(l3, l1) and torch.equal(sta._skip_list, one._skip_list)
        assert seen == [(i, a * bm, min(S, (a + n) * bm)) for i, (a, n) in enumerate(windows)]
    assert one.get_skip_fraction() > 0.02


def test_q_tile_window_leaves_other_rows_untouched_and_rejects_bad_windows():
    L = _L()
    from liteattention_amd.flash_attn_interface import mha_fwd
    B, S, H, D = 1, 1024, 2, 128
    bm, _ = L.get_tile_sizes(D, 2)
    Qt = S // bm
    g = torch.Generator().manual_seed(3)
    q, k, v = [torch.randn(B, S, H, D, generator=g).bfloat16().cuda() for _ in range(3)]
    full = L.flash_attn_func(q, k, v)
    out = torch.full_like(full, 7.0)
    u = 2 if os.environ.get("LA_VOTE", "").startswith("half") else 1        # LA_FLAG_HALF_VOTE: windows are pairs of 128-row q-tiles
    mha_fwd(q, k, v, out=out, _q_windows=[(u, u)])
    assert torch.equal(out[:, u * bm:2 * u * bm], full[:, u * bm:2 * u * bm])
    assert (out[:, :u * bm] == 7.0).all() and (out[:, 2 * u * bm:] == 7.0).all()
    for bad in ([(0, Qt + 1)], [(Qt, 1)], [(-1, 1)], [(0, 0)]):
        with pytest.raises(RuntimeError):
            mha_fwd(q, k, v, _q_windows=bad)


# ------------------------------------------------------------------------------------------ head dims between instantiations
@pytest.mark.parametrize("D,dtype", [(96, "bf16"), (40, "bf16"), (72, "bf16"), (96, "fp8"), (256, "bf16"), (192, "bf16"), (160, "bf16")])
def test_other_head_dims_run_on_the_next_instantiated_kernel(D, dtype):
    """head_dim 96 (instantiated by the reference, hopper/setup.py:58) and other multiples of 8 below 128 run the next
    kernel up on zero-padded operands; results must match the oracle at the ORIGINAL head_dim (scale D^-0.5) and the lists
    use the serving kernel's tiles."""
    L, orc = _L(), _orc()
    from liteattention_amd.flash_attn_interface import kernel_head_dim
    es = 1 if dtype == "fp8" else 2
    B, S, H = 1, 700, 2
    bm, bn = L.get_tile_sizes(D, es)
    assert (bm, bn) == L.get_tile_sizes(kernel_head_dim(D, es), es)
    q, k, v = structured_qkv(B, S, H, D, seed=D)
    if dtype == "fp8":
        q, k, v = [x.float().to(F8) for x in (q, k, v)]
    att = L.LiteAttention(threshold=-3.0, max_batch_size=B)
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    margins = torch.empty(B, H, Qt, Kt)
    for step in range(2):
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        assert out.shape == (B, S, H, D)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, thr=-3.0,
                                           margins=margins, p_round=fp8_p_round() if dtype == "fp8" else True)
        tol = (0.05 * o_ref.abs().max().item() + 2e-2) if dtype == "fp8" else _oracle_tol(o_ref)
        assert (out.float().cpu() - o_ref).abs().max().item() <= tol
        assert (lse.cpu() - lse_ref).abs().max().item() <= (fp8_lse_tol() if dtype == "fp8" else 1e-3)
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, -3.0, B)
        assert bad == 0


# ------------------------------------------------------------------------------------------ malformed lists
@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("bf16", 64), ("fp8", 128)])
def test_malformed_read_lists_are_memory_safe(dtype, D):
    """Caller-owned lists can hold anything. Indices are clamped to [0, Kt), the number of walked tiles to Kt, the first
    tile of the first range is always walked (as the reference's prologue does, mainloop...:1614-1660): garbage lists must
    neither fault nor hang, and an inverted first range behaves as in the oracle."""
    L, orc = _L(), _orc()
    es = 1 if dtype == "fp8" else 2
    B, S, H = 1, 1000, 2
    bm, bn = L.get_tile_sizes(D, es)
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    q, k, v = structured_qkv(B, S, H, D, seed=5)
    if dtype == "fp8":
        q, k, v = [x.float().to(F8) for x in (q, k, v)]
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    g = torch.Generator().manual_seed(99)
    for trial in range(6):
        lists = torch.randint(-2 ** 31, 2 ** 31 - 1, (2, B, H, Qt, Kt + 1), generator=g, dtype=torch.int64).to(torch.int32)
        if trial % 2:
            lists = lists % 23 - 3                             # small values: plausible-looking but inconsistent rows
        lists = lists.cuda()
        out, lse = L.flash_attn_func(qd, kd, vd, attn_read_list=lists[0], attn_write_list=lists[1], thr=-3.0,
                                     return_softmax_lse=True)
        torch.cuda.synchronize()
        assert out.shape == qd.shape and bool(torch.isfinite(out.float()).all())
        w = lists[1]
        assert bool((w[..., 0] >= 0).all()) and bool((w[..., 0] <= Kt).all())      # the writer stays inside its row
    # inverted first range [2, 3, 7]: start < end -> exactly tile 3 is computed, by the kernel and by the oracle
    rows = torch.zeros(2, B, H, Qt, Kt + 1, dtype=torch.int32)
    rows[..., 0], rows[..., 1], rows[..., 2] = 2, 3, 7
    out = L.flash_attn_func(qd, kd, vd, attn_read_list=rows[0].cuda(), attn_write_list=rows[1].cuda(), thr=-3.0)
    wr = torch.zeros_like(rows[1])
    o_ref, _, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rows[0], write_list=wr, thr=-3.0,
                                       p_round=fp8_p_round() if dtype == "fp8" else True)
    assert n_tiles == B * H * Qt
    tol = (0.05 * o_ref.abs().max().item() + 2e-2) if dtype == "fp8" else _oracle_tol(o_ref)
    assert (out.float().cpu() - o_ref).abs().max().item() <= tol


# ------------------------------------------------------------------------------------------ maximum key length
@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_longest_supported_key_sequence_and_the_typed_error_beyond_it(dtype):
    """The expanded read list of a workgroup and the tile-address table of its walk live in LDS beside the K/V rings (20.25 bytes
    per key tile): ~4 800 key tiles (Sk ~ 310 000) fit. A walk of 4 500 tiles must match a torch fp32 reference; a tile count past
    the limit must raise the typed error, not launch."""
    L = _L()
    D, H, Sq = 128, 1, 300
    Sk = 4500 * 64 - 17                                   # ragged last tile
    g = torch.Generator().manual_seed(42)
    q = torch.randn(1, Sq, H, D, generator=g)
    k = torch.randn(1, Sk, H, D, generator=g)
    v = torch.randn(1, Sk, H, D, generator=g)
    if dtype == "fp8":
        q, k, v = [x.to(F8) for x in (q, k, v)]
    else:
        q, k, v = [x.bfloat16() for x in (q, k, v)]
    att = L.LiteAttention(threshold=-40.0, max_batch_size=1)           # lists walked, nothing dropped
    out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    dense, lse_d = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert torch.equal(out, dense) and torch.equal(lse, lse_d)
    qf, kf, vf = [x.float().cuda()[0, :, 0] for x in (q, k, v)]
    sc = qf @ kf.T / D ** 0.5
    ref = torch.softmax(sc, -1) @ vf
    tol = 0.05 * ref.abs().max().item() + 2e-2 if dtype == "fp8" else 2.0 ** -7 * ref.abs().max().item() + 1e-3
    assert (out.float()[0, :, 0] - ref).abs().max().item() <= tol
    # fp8 default: row sums of the encoded P. Over 288 k keys the noise averages out but its BIAS does not (log-linear encoding: about
    # -3e-4; hardware rounding under LA_FLAG_FP8_MFMA_ROWSUM: P is log-uniform inside an e4m3 rounding interval, so round-to-nearest loses
    # (step / value)^2 / 12 ~ 7e-4 of the sum, every row low); the default form (fp32 row sums) has neither (tests/test_gpu_fp8.py runs all three)
    assert (lse[0, 0] - torch.logsumexp(sc, -1)).abs().max().item() <= (2.5e-3 if dtype == "fp8" else 1e-3)
    assert att.get_skip_fraction() == 0.0
    # past the limit: skip lists raise the typed error, nothing is launched; a DENSE call (every dtype, round 6) is cut into runs inside la_fwd and
    # merged: a strided `out` (a head slice of a wider tensor) is written in place by the merge. (e4m3 walks fit ~6 400 tiles, bf16 ~4 800.)
    Sk2 = (7000 if dtype == "fp8" else 6000) * 64 - 3
    k2 = torch.randn(1, Sk2, H, D, generator=g)
    v2 = torch.randn(1, Sk2, H, D, generator=g)
    k2, v2 = ([x.to(F8) for x in (k2, v2)] if dtype == "fp8" else [x.bfloat16() for x in (k2, v2)])
    from liteattention_amd.flash_attn_interface import mha_fwd
    wide = torch.full((1, Sq, 3, D), 7.0, dtype=torch.bfloat16, device="cuda")
    o2, l2, *_ = mha_fwd(q.cuda(), k2.cuda(), v2.cuda(), out=wide[:, :, 1:2])
    sc2 = qf @ k2.float().cuda()[0, :, 0].T / D ** 0.5
    ref2 = torch.softmax(sc2, -1) @ v2.float().cuda()[0, :, 0]
    tol2 = 0.05 * ref2.abs().max().item() + 2e-3 if dtype == "fp8" else 2.0 ** -7 * ref2.abs().max().item() + 1e-3
    assert (wide[0, :, 1].float() - ref2).abs().max().item() <= tol2
    assert (l2[0, 0] - torch.logsumexp(sc2, -1)).abs().max().item() <= (2.5e-3 if dtype == "fp8" else 1e-3)
    assert (wide[:, :, 0] == 7.0).all() and (wide[:, :, 2] == 7.0).all()
    bm, bn = L.get_tile_sizes(D, q.element_size())
    lists = torch.zeros(2, 1, H, -(-Sq // bm), -(-Sk2 // bn) + 1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="too long"):
        L.flash_attn_func(q.cuda(), k2.cuda(), v2.cuda(), attn_read_list=lists[0], attn_write_list=lists[1])


# ------------------------------------------------------------------------------------------ ticket queues cover every item
@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("bf16", 64), ("fp8", 128)])
@pytest.mark.parametrize("B,H,S", [(1, 5, 2400), (3, 3, 700), (1, 1, 9000), (2, 7, 260)])
def test_dynamic_work_distribution_computes_every_item_exactly(dtype, D, B, H, S):
    """Per-XCD ticket queues + stealing must hand out every (batch, head, q-tile) once: head groups of 8 with a short last
    group (B*H = 5, 9, 1, 14), chunks with a short last chunk, fewer items than workgroups. `out` is pre-filled with NaN
    (a fresh allocation could still hold an earlier, identical result), and the result must equal the static map's bit for bit."""
    L = _L()
    from liteattention_amd.flash_attn_interface import mha_fwd
    es = 1 if dtype == "fp8" else 2
    bm, bn = L.get_tile_sizes(D, es)
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    g = torch.Generator().manual_seed(B * 100 + H)
    q, k, v = [torch.randn(B, S, H, D, generator=g) for _ in range(3)]
    q, k, v = [(x.to(F8) if dtype == "fp8" else x.bfloat16()).cuda() for x in (q, k, v)]
    lists = L.LiteAttention.init_skip_list(B, S, H, D, False, q.dtype, "cuda")
    res = []
    for static in (False, True):
        out = torch.full((B, S, H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        lists[1].zero_()
        mha_fwd(q, k, v, out=out, attn_read_list=lists[0], attn_write_list=lists[1], thr=-30.0, _static_sched=static)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(out.float()).any()), "an item was never computed"
        assert bool((lists[1][..., 0] == 2).all()) and bool((lists[1][..., 1] == Kt - 1).all())      # every row was written
        res.append(out)
    assert torch.equal(res[0], res[1])


def test_reference_profile_script_head_dims():
    """/root/reference/profile_lite_attention.py: LiteAttention on (1, 10000, 4, d) bf16 for d in 32, 64, 96, 128, 192, 256 with
    threshold +2 (a tile survives only if it raises some row's running max by more than 2^2), two calls each, then prints
    `_skip_list.shape`. Every head_dim must run; after the calls almost everything is skipped, and every row still starts
    with the tile that can never be dropped."""
    L = _L()
    torch.manual_seed(0)
    for head_dim in [32, 64, 96, 128, 192, 256]:
        attn = L.LiteAttention()
        attn.threshold = float(2)
        for i in range(2):
            q, k, v = [torch.randn(1, 2 * 5000, 4, head_dim, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
            output = attn(q, k, v)
        torch.cuda.synchronize()
        bm, bn = L.get_tile_sizes(head_dim, 2)
        Qt, Kt = math.ceil(10000 / bm), math.ceil(10000 / bn)
        assert tuple(attn._skip_list.shape) == (2, 1, 4, Qt, Kt + 1) and output.shape == q.shape      # the batch seen (reference: max_batch_size = 4)
        cur = attn.current_read_list()[:1]
        assert bool((cur[..., 0] >= 2).all()) and bool((cur[..., 1] == Kt - 1).all())
        assert attn.get_skip_fraction(batch=1) > 0.9
        assert bool(torch.isfinite(output.float()).all())
