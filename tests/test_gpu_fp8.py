"""GPU: fp8 (e4m3) path — BASELINE.json configs[4] — against the reference outputs and the CPU oracle.

Tolerance restated for fp8: dense O vs the reference's fp32 `out_ref` uses the reference's own rule with `out_pt`
computed with P cast to e4m3 (hopper/tests/test_flash_attn.py:253,296): measured bound ~0.17-0.26 absolute on
these cases (|O| <= ~2). Against the tiled oracle, which rounds P to e4m3 exactly like the kernel, the bound is
0.05 * max|O| + 2e-2: v_exp_f32 and libm exp2f differ in the last fp32 bits, so a P that sits on an e4m3 rounding
boundary can land one e4m3 step (6-12 % of that P) apart; with few keys one P carries O(1) of a row's weight.
Still 3-5x tighter than the reference's own fp8 rule. The same bound holds for the default block-scaled log-linear byte encoding of P (round 3: per element
within [-8.0 %, +6.5 %] of P against +-6.25 % for the hardware rounding, include/lite_attention_amd.h LA_FLAG_FP8_ENCODED_P), checked against the oracle's
restatement of that encoding (p_round="fp8_lin", itself pinned against the reference-generated outputs in tests/test_oracle.py). LSE: `helpers.fp8_lse_tol()`
per form of P. Every test of this module runs in all three forms."""
import math

import pytest
import torch

from helpers import FP8_CASES, fp8_lse_tol, fp8_lse_tol_vs_exact, fp8_rows_off_grid, load_dense_case, ref_tolerance, structured_qkv, fp8_p_round

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def _tiles():
    import liteattention_amd as L
    return L.get_tile_sizes(128, 1)           # (256, 64) x64 kernel; (128, 64) with LA_FP8_KERNEL=v1


BM, BN = _tiles()


@pytest.fixture(params=["encoded", "exp", "exact"], autouse=True)
def p_mode(request, monkeypatch):
    """The three forms of P in the fp8 kernel: LA_FP8_P=encoded -> LA_FLAG_FP8_ENCODED_P (log-linear byte encoding, row sums of the encoded
    P from the matrix pipe), LA_FP8_P=mfma_rowsum -> LA_FLAG_FP8_MFMA_ROWSUM (v_exp_f32 + hardware e4m3 rounding, row sums of the rounded P
    from the matrix pipe) and the DEFAULT (that, with the fp32 sum of the un-rounded P on the vector unit: the reference's form)."""
    monkeypatch.delenv("LA_FP8_P", raising=False)
    if request.param == "encoded":
        monkeypatch.setenv("LA_FP8_P", "encoded")
    elif request.param == "exp":
        monkeypatch.setenv("LA_FP8_P", "mfma_rowsum")
    return request.param


def _tol(o):
    return 0.05 * o.abs().max().item() + 2e-2


@pytest.mark.parametrize("name", FP8_CASES)
def test_fp8_dense_matches_reference_outputs(name):
    import liteattention_amd as L
    from oracle import oracle as orc
    c = load_dense_case(name)
    q, k, v = [x.to(F8).cuda() for x in (c["q"], c["k"], c["v"])]
    qd, kd, vd = [c[n].cuda() for n in ("q_descale", "k_descale", "v_descale")]
    out, lse = L.flash_attn_func(q, k, v, q_descale=qd, k_descale=kd, v_descale=vd, return_softmax_lse=True)
    assert out.dtype == torch.bfloat16 and out.shape == q.shape                      # bf16 out, flash_api.cpp:859
    err = (out.float().cpu() - c["out_ref"]).abs().max().item()
    assert err <= ref_tolerance(c["out_ref"], c["pt_maxerr"]), (err, ref_tolerance(c["out_ref"], c["pt_maxerr"]))
    assert (lse.cpu() - c["lse_ref"]).abs().max().item() <= fp8_lse_tol()
    o8, lse8, _ = orc.qkskip_fwd(c["q"], c["k"], c["v"], block_m=BM, block_n=BN, p_round=fp8_p_round(),
                                 q_descale=c["q_descale"], k_descale=c["k_descale"], v_descale=c["v_descale"])
    assert (out.float().cpu() - o8).abs().max().item() <= _tol(o8)
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()


@pytest.mark.parametrize("shape", [(1, 17, 1, 17), (2, 129, 3, 65), (1, 1000, 2, 1250), (1, 128, 1, 4224)])
def test_fp8_ragged_shapes_no_descale(shape):
    import liteattention_amd as L
    from oracle import oracle as orc
    B, Sq, H, Sk = shape
    g = torch.Generator().manual_seed(Sq * 7 + Sk)
    q = torch.randn(B, Sq, H, 128, generator=g).to(F8)
    k = torch.randn(B, Sk, H, 128, generator=g).to(F8)
    v = torch.randn(B, Sk, H, 128, generator=g).to(F8)
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    o8, lse8, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, p_round=fp8_p_round())
    assert (out.float().cpu() - o8).abs().max().item() <= _tol(o8)
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()


def test_fp8_skip_lists_match_oracle_over_steps():
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    B, S, H, thr = 1, 1536, 2, -3.0
    Qt, Kt = S // BM, S // BN
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    md_row = orc.expand_must_do_ref([0, 0], BN, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    listed = []
    # descales through LiteAttention.__call__ (keyword-only extension): per (batch, K/V head)
    qd, kd, vd = torch.tensor([[0.7, 1.3]]), torch.tensor([[1.1, 0.9]]), torch.tensor([[0.5, 1.7]])
    for step in range(4):
        q, k, v = [x.to(F8) for x in structured_qkv(B, S, H, 128, seed=300, alpha=9.0, dtype=torch.float32)]
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, q_descale=qd.cuda(), k_descale=kd.cuda(),
                       v_descale=vd.cuda())
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc,
                                           must_do_list=md_row, thr=thr, margins=margins, p_round=fp8_p_round(),
                                           q_descale=qd, k_descale=kd, v_descale=vd)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0
        listed.append(orc.listed_tiles(wr[:B]))
    assert listed[-1] < 0.95 * B * H * Qt * Kt and listed == sorted(listed, reverse=True)


def test_fp8_errors():
    import liteattention_amd as L
    q = torch.randn(1, 256, 2, 128, device="cuda").to(F8)
    with pytest.raises(RuntimeError, match="q_descale"):
        L.flash_attn_func(q, q, q, q_descale=torch.ones(2, 2, device="cuda"))             # wrong shape
    qb = torch.randn(1, 256, 2, 128, device="cuda").bfloat16()
    with pytest.raises(RuntimeError, match="only supported for fp8"):
        L.flash_attn_func(qb, qb, qb, q_descale=torch.ones(1, 2, device="cuda"))
    q272 = torch.randn(1, 256, 2, 272, device="cuda").to(F8)
    with pytest.raises(RuntimeError, match="head_size"):
        L.flash_attn_func(q272, q272, q272)                                                 # head_dim > 256: no kernel for any type
    q72 = torch.randn(1, 256, 2, 72, device="cuda").bfloat16().view(torch.int16)[..., :72].view(torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        L.flash_attn_func(q72.to(F8), q72.to(F8), q72.to(F8))                               # fp8: head_size % 16 (:854-856)


@pytest.mark.parametrize("gain", [3.0, 8.0])
def test_fp8_running_max_that_grows_late_in_the_walk(gain):
    """The fp8 x64 kernel keeps O and l relative to a reference max that follows the true running max only when it has
    grown by more than tau = 2 (log2 units), with P offset 2^(8 - tau): keys walked LAST score far higher here, so every row's
    max keeps growing through the walk — the O^T rescale round trip fires repeatedly and P sits at the top of its e4m3 range.
    Lists must match the oracle (votes use the true running max), outputs stay inside the fp8 tolerance, nothing saturates."""
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    B, S, H = 1, 1536, 2
    g = torch.Generator().manual_seed(91)
    q, k, v = [torch.randn(B, S, H, 128, generator=g) for _ in range(3)]
    k = k * torch.linspace(gain, 1.0, S).view(1, S, 1, 1)                  # early keys (walked last) up to `gain` x larger
    q, k, v = [x.to(F8) for x in (q, k, v)]
    o8, lse8, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, p_round=fp8_p_round())
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert bool(torch.isfinite(out.float()).all())
    assert (out.float().cpu() - o8).abs().max().item() <= _tol(o8)
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()
    # two independent bounds on the LSE: against the EXACT value (the encoding's stated bound), and - the oracle being on the kernel's
    # own grid - all but a fraction of a percent of the rows agree with the same-form oracle to 0.01 (peaked rows here: a byte of the
    # dominant key is decided by the last bits of S on a handful of them)
    _, lse_exact, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, p_round=False)
    assert (lse.cpu() - lse_exact).abs().max().item() <= fp8_lse_tol_vs_exact()
    if fp8_p_round() == "fp8_lin":                   # (the exact forms: the kernel's P is relative to ITS lazy reference with tau = 2, the oracle's to the true maximum)
        assert fp8_rows_off_grid(lse.cpu(), lse8) <= 0.01
    Qt, Kt = -(-S // BM), -(-S // BN)
    att = L.LiteAttention(threshold=-1.0, max_batch_size=B)
    margins = torch.empty(B, H, Qt, Kt)
    for _ in range(2):
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc, thr=-1.0,
                                           margins=margins, p_round=fp8_p_round())
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, -1.0, B)
        assert bad == 0


@pytest.mark.parametrize("D", [192, 256, 160])
def test_fp8_above_head_dim_128(D):
    """e4m3 above head_dim 128 (round 6: native bodies with one 32-row q-block per wave, q-tile 128 - until then the bf16 kernels on up-converted
    operands; 160 runs zero-padded on the 192 body). Dense with GQA + descales, then three steps of lists with descales, against the oracle in the
    same form of P with these bodies' tiles. More cases: tests/test_gpu_fp8_head_dims.py."""
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    bm, bn = L.get_tile_sizes(D, 1)
    assert (bm, bn) == (128, 64)
    g = torch.Generator().manual_seed(D)
    B, Sq, Sk, H, Hk = 2, 300, 1000, 4, 2
    q, k, v = torch.randn(B, Sq, H, D, generator=g).to(F8), torch.randn(B, Sk, Hk, D, generator=g).to(F8), torch.randn(B, Sk, Hk, D, generator=g).to(F8)
    qd, kd, vd = [0.5 + torch.rand(B, Hk, generator=g) for _ in range(3)]
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), q_descale=qd.cuda(), k_descale=kd.cuda(), v_descale=vd.cuda(), return_softmax_lse=True)
    assert out.dtype == torch.bfloat16
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=fp8_p_round(), q_descale=qd, k_descale=kd, v_descale=vd)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()
    S, thr = 1536, -3.0
    Qt, Kt = -(-S // bm), -(-S // bn)
    att = L.LiteAttention(threshold=thr, max_batch_size=1)
    md_row = orc.expand_must_do_ref([0, 0], bn, Kt + 1)
    margins = torch.empty(1, 2, Qt, Kt)
    qd2, kd2, vd2 = torch.tensor([[2.0, 0.5]]), torch.tensor([[0.5, 1.0]]), torch.tensor([[4.0, 0.25]])
    for step in range(3):
        q, k, v = [x.to(F8) for x in structured_qkv(1, S, 2, D, seed=310 + D, alpha=9.0 - step, dtype=torch.float32)]
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, q_descale=qd2.cuda(), k_descale=kd2.cuda(), v_descale=vd2.cuda())
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, must_do_list=md_row, thr=thr,
                                           margins=margins, p_round=fp8_p_round(), q_descale=qd2, k_descale=kd2, v_descale=vd2)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, 1)
        assert bad == 0


def test_the_kernels_bytes_of_p_are_the_oracles_on_exact_scores(p_mode):
    """"make the oracle's fp8_lin encoder and the kernel's bit-identical". They cannot be on arbitrary inputs - the
    scores S themselves differ in their last bits between the MFMA's accumulation and any CPU order, and a score on a byte boundary then
    lands one byte (2^(1/8) = 9 % of that weight) apart. On scores that are EXACT in fp32 whatever the order (small integers: q, k in
    {-2..2}, 128 products) everything downstream is the same fp32 operations in the same order on both sides (oracle p_round 4 with the
    lazy reference maximum, round 5: set_nms / mx_ops / the byte convert of gen_fwd_x64_fp8.py restated operation by operation), so every
    byte of P~ must be the same. The LSE shows it: it is fp32, ln(sum of P~) - one flipped byte of a key carrying weight w moves it by
    0.09 w, while the order of the fp32 sums moves it by ~1e-5 at LSE ~ 20. Bound 1e-4: no key above 1e-3 of its row's mass has another byte. Several
    tiles, growing maxima (the block scales do the work), GQA, a ragged last tile. (The exact forms are held to the same bound: there the
    hardware's v_exp_f32 and libm's exp2f differ in the last bit, which the e4m3 rounding can turn into a step: bound 0.07 w instead.)"""
    import liteattention_amd as L
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(11)
    B, Sq, Sk, H, Hk, D = 1, 300, 1000, 4, 2, 128
    q = torch.randint(-2, 3, (B, Sq, H, D), generator=g).float()
    k = torch.randint(-2, 3, (B, Sk, Hk, D), generator=g).float()
    k[:, :200] *= 2.0                                                     # walked last (descending walk): the maxima grow along the walk
    v = torch.randn(B, Sk, Hk, D, generator=g).to(F8).float()
    q8, k8, v8 = q.to(F8), k.to(F8), v.to(F8)
    assert torch.equal(q8.float(), q) and torch.equal(k8.float(), k)      # exactly representable
    scale = 0.03
    out, lse = L.flash_attn_func(q8.cuda(), k8.cuda(), v8.cuda(), softmax_scale=scale, return_softmax_lse=True)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, p_round=fp8_p_round(), softmax_scale=scale)
    err_l = (lse.cpu() - lse_ref).abs().max().item()
    if fp8_p_round() == "fp8_lin":
        assert err_l <= 1e-4, err_l                                       # the same bytes (measured 3e-5: the fp32 product m_ref c ln 2 at LSE ~ 20)
        assert (out.float().cpu() - o_ref).abs().max().item() <= 2.0 ** -7 * o_ref.abs().max().item() + 1e-4     # + the bf16 rounding of O
    else:
        assert err_l <= fp8_lse_tol()
