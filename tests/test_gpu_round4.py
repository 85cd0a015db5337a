"""GPU: round-4 additions at the boundary: ``la_blockmask_to_lists`` (device kernel, bit-exact against the pure-Python
``blockmask_to_rows``), ``lite_attention::fwd_combine`` (the reference's op contract), ``la_device_slots``."""
import ctypes

import pytest
import torch

import liteattention_amd as L
from liteattention_amd import _cabi, compat
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rows_ref(mask2d, kv=None):
    """The oracle's rows (oracle.blockmask_rows_ref: pure Python, one tile at a time), cross-checked against the product's own pure-Python
    ``compat.blockmask_to_rows`` wherever that one is defined (rows that keep a tile)."""
    ref = orc.blockmask_rows_ref(mask2d != 0, kv)
    m = (mask2d != 0).clone()
    if kv is not None:
        m[:, kv:] = False
    kt = m.shape[1]
    for i in range(m.shape[0]):
        if bool(m[i].any()):
            r = compat.blockmask_to_rows(m[i:i + 1])[0][: kt + 1]
            assert ref[i, : len(r)].tolist() == r and r[0] == int(ref[i, 0])
    return ref


@pytest.mark.parametrize("qt,kt,p", [(5, 1, 0.5), (7, 2, 0.5), (9, 3, 0.6), (33, 63, 0.5), (17, 64, 0.3), (12, 65, 0.7), (40, 130, 0.5),
                                     (296, 1182, 0.56), (3, 257, 1.0), (4, 200, 0.02), (2, 4200, 0.5)])     # 4200: past the LDS-staged form
def test_blockmask_kernel_rows_equal_the_python_rows_bit_for_bit(qt, kt, p):
    g = torch.Generator().manual_seed(qt * 1000 + kt)
    mask = torch.rand(qt, kt, generator=g) < p
    ref = _rows_ref(mask)
    n_empty = int((~mask.any(-1)).sum())
    got = compat.blockmask_to_lists(mask.to(DEV), validate=False)
    assert got.dtype == torch.int32 and tuple(got.shape) == (qt, kt + 1)
    assert torch.equal(got.cpu(), ref)
    if n_empty:
        with pytest.raises(ValueError):
            compat.blockmask_to_lists(mask.to(DEV), validate=True)
    # alternating mask: the maximum number of runs (the last end may fall behind the row: counted, not stored)
    alt = (torch.arange(kt)[None, :] + torch.arange(qt)[:, None]) % 2 == 0
    assert torch.equal(compat.blockmask_to_lists(alt.to(DEV), validate=False).cpu(), _rows_ref(alt))
    # the host-side tensor-op form says the same
    assert torch.equal(compat.blockmask_to_lists(mask, validate=False), ref)


def test_blockmask_kernel_broadcast_strides_valid_counts_and_raw_cabi():
    B, H, qt, kt = 3, 4, 11, 37
    g = torch.Generator().manual_seed(5)
    mask = torch.rand(qt, kt, generator=g) < 0.5
    mask[:, 0] = True                                           # every row keeps a tile for every clip >= 1
    kv = torch.tensor([37, 5, 1], dtype=torch.int32)
    qv = torch.tensor([11, 3, 0], dtype=torch.int32)
    got = compat.blockmask_to_lists(mask.to(DEV), k_tiles_valid=kv.to(DEV), q_tiles_valid=qv.to(DEV), batch=B, heads=H)
    assert tuple(got.shape) == (B, H, qt, kt + 1)
    for b in range(B):
        ref = _rows_ref(mask, int(kv[b]))
        full = _rows_ref(torch.ones(qt, kt, dtype=torch.bool), int(kv[b]))
        ref[int(qv[b]):] = full[int(qv[b]):]                    # q-tiles past the sequence's end: the whole corner
        for h in range(H):
            assert torch.equal(got[b, h].cpu(), ref), (b, h)
    # per-(batch, head) masks through the raw C-ABI, non-bool mask values, uint8 input
    m4 = (torch.rand(B, H, qt, kt, generator=g) < 0.4).to(torch.uint8) * 7
    m4[..., -1] = 1
    d = m4.to(DEV)
    lists = torch.full((B, H, qt, kt + 1), -1, dtype=torch.int32, device=DEV)
    empty = torch.full((1,), 99, dtype=torch.int32, device=DEV)
    rc = _cabi.load().la_blockmask_to_lists(d.data_ptr(), d.stride(0), d.stride(1), B, H, qt, kt, None, None, lists.data_ptr(),
                                            empty.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == _cabi.LA_OK and int(empty.item()) == 0
    for b in range(B):
        for h in range(H):
            assert torch.equal(lists[b, h].cpu(), _rows_ref(m4[b, h] != 0))
    assert torch.equal(compat.blockmask_to_lists(d), lists)
    lib = _cabi.load()
    assert lib.la_blockmask_to_lists(None, 0, 0, 1, 1, 1, 1, None, None, lists.data_ptr(), None, None) == _cabi.LA_ERR_NULL_ARG
    assert lib.la_blockmask_to_lists(d.data_ptr(), 0, 0, 0, 1, 1, 1, None, None, lists.data_ptr(), None, None) == _cabi.LA_ERR_SHAPE
    assert lib.la_blockmask_to_lists(d.data_ptr(), -1, 0, 1, 1, 1, 1, None, None, lists.data_ptr(), None, None) == _cabi.LA_ERR_STRIDE


def test_blockmask_lists_drive_the_kernel_like_a_masked_dense_attention():
    """End to end: mask -> la_blockmask_to_lists -> la_fwd == the oracle walking the Python rows."""
    B, S, H, D = 1, 1500, 2, 128
    bm, bn = L.get_tile_sizes(D, 2)
    qt, kt = -(-S // bm), -(-S // bn)
    g = torch.Generator().manual_seed(11)
    q, k, v = [torch.randn(B, S, H, D, generator=g).bfloat16() for _ in range(3)]
    mask = torch.rand(qt, kt, generator=g) < 0.5
    mask[:, -1] = True
    out, lse = L.flash_blocksparse_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), return_softmax_lse=True)
    rows = _rows_ref(mask)[None, None].expand(B, H, qt, kt + 1).contiguous()
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rows, write_list=torch.zeros_like(rows), thr=float("-inf"))
    assert (out.float().cpu() - o_ref).abs().max().item() <= 2.0 ** -8 * o_ref.abs().max().item() + 1e-3
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("num_splits,seqlen,d", [(1, 1, 64), (2, 3, 128), (5, 113, 96), (17, 64, 256), (3, 108, 512), (4, 33, 59)])
def test_fwd_combine_op_follows_the_reference_test(num_splits, seqlen, d, dtype):
    """The reference's own test of the op (hopper/tests/test_flash_attn.py:1190-1229): non-contiguous partials, -inf splits, the
    transposed LSE layout, its tolerance rule. ``flash_attn_combine`` is the same op call, bit for bit."""
    torch.random.manual_seed(1)
    batch_size, nheads = 5, 16
    out_partial = torch.randn(num_splits * 2, batch_size, nheads, seqlen, d, device=DEV, dtype=torch.float32).transpose(2, 3)[:num_splits]
    lse_partial = torch.randn(num_splits, batch_size, nheads * 2, seqlen, device=DEV, dtype=torch.float32).transpose(-1, -2)[:, :, :, :nheads]
    lse_partial[num_splits // 2:, :batch_size // 3] = -float("inf")
    out, lse = torch.ops.lite_attention.fwd_combine(out_partial, lse_partial, None, dtype)
    out_ref, lse_ref = orc.attention_combine_ref(out_partial.cpu(), lse_partial.cpu())
    out_pt = out_ref.to(dtype)
    assert tuple(out.shape) == (batch_size, seqlen, nheads, d) and out.dtype == dtype and tuple(lse.shape) == (batch_size, seqlen, nheads)
    assert torch.allclose(lse.cpu(), lse_ref, atol=1e-5, rtol=1e-5)
    assert ((out.cpu() - out_ref).abs().max().item() <= 2 * (out_pt - out_ref).abs().max().item()) or torch.allclose(out.cpu(), out_pt, atol=1e-5, rtol=1e-5)
    out2, lse2 = L.flash_attn_combine(out_partial, lse_partial, out_dtype=dtype)
    assert torch.equal(out2, out) and torch.equal(lse2, lse)
    # the layout flash_attn_func returns (num_splits, batch, nheads, seqlen) gives the same numbers, LSE in that layout
    out3, lse3 = L.flash_attn_combine(out_partial, lse_partial.transpose(-1, -2).contiguous(), out_dtype=dtype)
    assert torch.equal(out3, out) and torch.equal(lse3, lse.transpose(1, 2))
    # out= is filled in place
    buf = torch.empty(batch_size, seqlen, nheads, d, device=DEV, dtype=dtype)
    out4, _ = torch.ops.lite_attention.fwd_combine(out_partial, lse_partial, buf, dtype)
    assert out4.data_ptr() == buf.data_ptr() and torch.equal(buf, out)


def test_fwd_combine_error_behaviour():
    op = torch.zeros(2, 1, 4, 2, 64, device=DEV)
    lp = torch.zeros(2, 1, 2, 4, device=DEV).transpose(-1, -2)
    with pytest.raises(RuntimeError, match="fp32"):
        torch.ops.lite_attention.fwd_combine(op, lp.double(), None, None)
    with pytest.raises(RuntimeError, match="seqlen dimension"):
        torch.ops.lite_attention.fwd_combine(op, torch.zeros(2, 1, 4, 2, device=DEV), None, None)
    with pytest.raises(RuntimeError, match="Output type"):
        torch.ops.lite_attention.fwd_combine(op, lp, None, torch.float64)
    with pytest.raises(RuntimeError, match="at most 256"):
        torch.ops.lite_attention.fwd_combine(torch.zeros(257, 1, 4, 2, 64, device=DEV), torch.zeros(257, 1, 2, 4, device=DEV).transpose(-1, -2), None, None)


def test_device_slots_come_from_the_device():
    cus, per = _cabi.device_slots(128, 2)
    assert cus == torch.cuda.get_device_properties(0).multi_processor_count and per == 1
    assert _cabi.device_slots(128, 1) == (cus, 1)
    assert _cabi.device_slots(128, 2, _cabi.LA_FLAG_KERNEL_128ROW) == (cus, 2) and _cabi.device_slots(256, 2, _cabi.LA_FLAG_KERNEL_128ROW) == (cus, 1)
    from liteattention_amd.parallel import HeadShardedLiteAttention   # noqa: F401  (plan_q_windows asks the library: test_distributed_cpu)


@pytest.mark.parametrize("D", [64, 96])
def test_fp8_packed_batch_below_head_dim_128_keeps_descales_and_returns_bf16(D):
    """the zero-padded recursion of the varlen path dropped q/k/v_descale and allocated `out` as e4m3. Per
    sequence the packed result must equal the fixed-length fp8 call (which pads the same way) bit for bit, with non-unit descales,
    and stay inside the fp8 bound against the oracle on the descaled operands."""
    from liteattention_amd.flash_attn_interface import mha_fwd
    F8 = torch.float8_e4m3fn
    H, Hk = 4, 2
    lens_q, lens_k = [300, 77, 0, 520], [410, 64, 30, 520]
    B = len(lens_q)
    g = torch.Generator().manual_seed(21)
    qs = [torch.randn(n, H, D, generator=g).to(F8) for n in lens_q]
    ks = [torch.randn(n, Hk, D, generator=g).to(F8) for n in lens_k]
    vs = [torch.randn(n, Hk, D, generator=g).to(F8) for n in lens_k]
    cat8 = lambda ts: torch.cat([t.view(torch.uint8) for t in ts]).view(F8)          # noqa: E731
    cu = lambda lens: torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)   # noqa: E731
    qd, kd, vd = [(0.5 + 1.5 * torch.rand(B, Hk, generator=g)).to(DEV) for _ in range(3)]
    o, lse, *_ = mha_fwd(cat8(qs).to(DEV), cat8(ks).to(DEV), cat8(vs).to(DEV), cu_seqlens_q=cu(lens_q), cu_seqlens_k=cu(lens_k),
                         max_seqlen_q=max(lens_q), max_seqlen_k=max(lens_k), q_descale=qd, k_descale=kd, v_descale=vd)
    assert o.dtype == torch.bfloat16 and tuple(o.shape) == (sum(lens_q), H, D) and tuple(lse.shape) == (H, sum(lens_q))
    r0 = 0
    for b in range(B):
        if lens_q[b] == 0:
            continue
        sq = slice(r0, r0 + lens_q[b])
        r0 += lens_q[b]
        o_b, lse_b, *_ = mha_fwd(qs[b][None].to(DEV), ks[b][None].to(DEV), vs[b][None].to(DEV), q_descale=qd[b:b + 1],
                                 k_descale=kd[b:b + 1], v_descale=vd[b:b + 1])
        assert torch.equal(o[sq], o_b[0]) and torch.equal(lse[:, sq], lse_b[0]), b
        # against the oracle on the descaled operands (softmax_scale = D^-0.5 of the ORIGINAL head dim)
        rep = H // Hk
        qf = qs[b].float() * (qd[b].cpu() * kd[b].cpu()).repeat_interleave(rep)[None, :, None]
        vf = vs[b].float() * vd[b].cpu()[None, :, None]
        ref, lse_ref = orc.attention_dense_ref(qf[None], ks[b].float()[None], vf[None])
        assert (o[sq].float().cpu() - ref[0]).abs().max().item() <= 0.05 * ref.abs().max().item() + 2e-2, b
        assert (lse[:, sq].cpu() - lse_ref[0]).abs().max().item() <= 0.09, b
    with pytest.raises(RuntimeError, match="bf16 for fp8"):
        mha_fwd(cat8(qs).to(DEV), cat8(ks).to(DEV), cat8(vs).to(DEV), out=torch.empty(sum(lens_q), H, D, dtype=F8, device=DEV).view(F8),
                cu_seqlens_q=cu(lens_q), cu_seqlens_k=cu(lens_k), max_seqlen_q=max(lens_q), max_seqlen_k=max(lens_k))


# ------------------------------------------------------------------------------ small parity fills
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config0_golden_through_the_head_dim_64_kernel(dtype):
    """BASELINE.json configs[0] (dense, 1 head, S = 2048, D = 64, fp32 eager on the CPU): the reference-generated ``out_ref`` of that
    case against the head_dim-64 kernel on the inputs cast to bf16 / fp16, under the reference's own rule
    (2 x the error of the same-dtype eager pass + fwd_atol, hopper/tests/test_flash_attn.py:266-296). The fixture's pt_maxerr is that of
    an fp32 pass; the same-dtype pass is recomputed here from the cast inputs."""
    from tests.helpers import load_dense_case
    c = load_dense_case("cfg0_fp32_s2048_d64")
    q, k, v = [x.to(dtype) for x in (c["q"], c["k"], c["v"])]
    out, lse = L.flash_attn_func(q.to(DEV), k.to(DEV), v.to(DEV), return_softmax_lse=True)
    assert out.dtype == dtype and tuple(out.shape) == (1, 2048, 1, 64)
    o_pt, _ = orc.attention_dense_ref(q, k, v, upcast=False, reorder_ops=True)
    o_up, lse_up = orc.attention_dense_ref(q, k, v)                  # fp32 math on the CAST inputs: what the kernel is asked to compute
    tol = 2 * (o_pt.float() - o_up.float()).abs().max().item() + 2 * (c["out_ref"] + 0.3 - 0.3 - c["out_ref"]).abs().max().item()
    assert (out.float().cpu() - o_up.float()).abs().max().item() <= tol
    # and against the fp32 fixture itself: the input cast is the only extra error (same bound with the cast pass measured vs the fixture)
    tol_fix = 2 * (o_pt.float() - c["out_ref"]).abs().max().item() + 2 * (c["out_ref"] + 0.3 - 0.3 - c["out_ref"]).abs().max().item()
    assert (out.float().cpu() - c["out_ref"]).abs().max().item() <= tol_fix
    assert (lse.cpu() - lse_up).abs().max().item() <= 1e-3


@pytest.mark.parametrize("D", [32, 64, 96, 128, 192, 256])
def test_reference_script_known_answers_at_every_head_dim(D):
    """/root/reference/test_lite_attention.py:7-93 loops over head dims 32 / 64 / 96 / 128 / 192 / 256 (32 runs zero-padded on the 64
    kernel here): K1 thr = +inf -> every write row is [2, Kt-1, Kt-2]; K2 + must_do over everything -> write == read; K3 thr = -inf ->
    write == read; K4 thr = 0: LSE against logsumexp (the script accepts 0.1; 1e-3 here). Fewer heads than the script's 32 (time)."""
    import math
    torch.manual_seed(0)
    q, k, v = [torch.randn(2, 5000, 4, D, device=DEV, dtype=torch.bfloat16) for _ in range(3)]
    bm, bn = L.LiteAttention.get_MN(D, 2)
    Kt = math.ceil(5000 / bn)
    attn = L.LiteAttention()
    attn.threshold = float("inf")
    attn(q, k, v)
    assert (attn._skip_list[1, :2, ..., 0] == 2).all()
    assert (attn._skip_list[1, :2, ..., 1] == Kt - 1).all() and (attn._skip_list[1, :2, ..., 2] == Kt - 2).all()
    attn = L.LiteAttention()
    attn.threshold = float("inf")
    attn(q, k, v, must_do_list=[k.shape[1] - 1, 0])
    assert (attn._skip_list[1] == attn._skip_list[0]).all()
    attn = L.LiteAttention()
    attn.threshold = float("-inf")
    attn(q, k, v)
    assert (attn._skip_list[1] == attn._skip_list[0]).all()
    attn = L.LiteAttention()
    attn.threshold = 0.0
    out, lse = attn(q, k, v, return_softmax_lse=True)
    qr, kr = q.permute(0, 2, 1, 3).float(), k.permute(0, 2, 1, 3).float()
    lse_ref = torch.logsumexp(torch.matmul(qr, kr.transpose(-2, -1)) / D ** 0.5, dim=-1)
    assert (lse_ref - lse).abs().max().item() < 1e-3


# ------------------------------------------------------------------------------ fp8 partial results merged by LSE
@pytest.mark.parametrize("form", ["seq_parallel_default", "encoded", "exact_exp", "exact_rowsum"])
def test_fp8_four_split_merge_against_the_fp8_oracle_on_the_full_keys(form, monkeypatch):
    """A ring-style / sequence-parallel caller: K/V in 4 shards, one fp8 forward per shard (SeqParallelLiteAttention, one skip state
    each), partial results merged by their LSE (flash_attn_combine; oracle hopper/tests/test_flash_attn.py:1178-1187) - against the
    fp8 oracle (the reference's arithmetic, p_round='fp8', softmax.h:85-87,275-296) on the FULL K/V, under the stated fp8 bound
    (0.05 max|O| + 2e-2), in all three forms of P. `seq_parallel_default` is what the class does by itself for e4m3 inputs with
    return_softmax_lse=True even inside a block that opted into the encoded form: it clears the fast-form flags, so the merged LSE is fp32-exact (1e-3); the two forms whose row sums are those of the
    8-bit P carry that noise into the merge weights: their merged LSE is held to the per-form bound of tests/helpers.py and their
    merged O to the same fp8 bound."""
    from liteattention_amd import _cabi as C
    from liteattention_amd.flash_attn_interface import fwd_flags
    monkeypatch.delenv("LA_FP8_P", raising=False)
    F8 = torch.float8_e4m3fn
    B, Sq, Sk, H, Hk, D, G = 1, 700, 2048, 4, 2, 128, 4
    g = torch.Generator().manual_seed(33)
    q = torch.randn(B, Sq, H, D, generator=g).to(F8)
    k = torch.randn(B, Sk, Hk, D, generator=g).to(F8)
    v = torch.randn(B, Sk, Hk, D, generator=g).to(F8)
    qd, kd, vd = [(0.5 + torch.rand(B, Hk, generator=g)) for _ in range(3)]
    sp = L.SeqParallelLiteAttention(G, threshold=-30.0, max_batch_size=B)
    flags = {"seq_parallel_default": C.LA_FLAG_FP8_ENCODED_P, "encoded": C.LA_FLAG_FP8_ENCODED_P, "exact_exp": C.LA_FLAG_FP8_MFMA_ROWSUM, "exact_rowsum": 0}[form]
    sp.exact_fp8_lse = form == "seq_parallel_default"
    outs, lses = [], []
    Sl = Sk // G
    for j in range(G):
        with fwd_flags(flags):
            o, lse = sp(q.to(DEV), k[:, j * Sl:(j + 1) * Sl].to(DEV), v[:, j * Sl:(j + 1) * Sl].to(DEV), j, return_softmax_lse=True,
                        q_descale=qd.to(DEV), k_descale=kd.to(DEV), v_descale=vd.to(DEV))
        outs.append(o)
        lses.append(lse)
    out, lse = L.flash_attn_combine(torch.stack(outs), torch.stack(lses))
    assert out.dtype == torch.bfloat16 and tuple(lse.shape) == (B, H, Sq)
    rep = H // Hk
    bm, bn = L.get_tile_sizes(D, 1)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q.float(), k.float(), v.float(), block_m=bm, block_n=bn, p_round="fp8",
                                       q_descale=qd, k_descale=kd, v_descale=vd)
    err = (out.float().cpu() - o_ref).abs().max().item()
    assert err <= 0.05 * o_ref.abs().max().item() + 2e-2, (form, err)
    lse_tol = {"seq_parallel_default": 1e-3, "exact_rowsum": 1e-3, "exact_exp": 0.0606, "encoded": 0.083}[form]
    lse_err = (lse.cpu() - lse_ref).abs().max().item()
    assert lse_err <= lse_tol, (form, lse_err)
    # the merge itself against the merge oracle on the SAME partials (attention_combine_ref)
    m_ref, ml_ref = orc.attention_combine_ref(torch.stack(outs).float().cpu(), torch.stack(lses).cpu().transpose(-1, -2))
    assert (out.float().cpu() - m_ref).abs().max().item() <= 2.0 ** -7 * m_ref.abs().max().item() + 1e-5
    assert torch.allclose(lse.cpu(), ml_ref.transpose(1, 2), atol=1e-5, rtol=1e-5)
    if form == "seq_parallel_default":        # and the long-row statistics: exact row sums leave no bias in the merged LSE
        assert (lse.cpu() - lse_ref).mean().abs().item() <= 1e-4


def test_windowed_launches_are_bit_identical_under_a_co_running_copy_stream():
    """the overlapped all-gather puts copy kernels of another stream beside the next q-tile window. With no second
    GPU, that is emulated by what an all-gather is on the device - copies on a side stream behind each window's rows, plus a bandwidth
    hog that never stops: outputs, LSE and write lists of the windowed dynamic and static-after-first forms stay bit-identical to the
    single undisturbed launch (tools/window_interference.py measures what it costs: profiles/r05_window_interference.md). The reference has
    one launch per call and no counterpart (flash_fwd_launch_template.h:359)."""
    from liteattention_amd.parallel import plan_q_windows
    torch.manual_seed(0)
    S, H, D = 8300, 6, 128
    g = torch.Generator(device=DEV).manual_seed(5)
    q, k, v = [torch.randn(1, S, H, D, device=DEV, generator=g).bfloat16() for _ in range(3)]
    bm, bn = L.get_tile_sizes(D, 2)

    def fresh():
        att = L.LiteAttention(threshold=-3.0, max_batch_size=1)
        att(q, k, v); att(q, k, v)                                   # two steps: a real (non-trivial) read list
        return att
    a0 = fresh()
    ref_o, ref_l = a0(q, k, v, return_softmax_lse=True)
    ref_lists = a0._skip_list.clone()
    side = torch.cuda.Stream()
    hog_src = torch.empty(64 << 20, dtype=torch.uint8, device=DEV); hog_dst = torch.empty_like(hog_src)
    peers = torch.empty_like(ref_o)
    for n, sched in ((3, False), (3, "after_first"), (5, "after_first")):
        att = fresh()
        from liteattention_amd.flash_attn_interface import q_tiles_per_item
        u = q_tiles_per_item(D, 2)                                    # LA_VOTE=half: a workgroup item is two 128-row q-tiles; windows hold whole items
        qt_all = -(-S // bm)
        w = [(b0 * u, min(c0 * u, qt_all - b0 * u)) for b0, c0 in plan_q_windows(-(-qt_all // u), H, n, slots=16)]   # small "machine": several rounds per window
        assert len(w) >= 2

        def hook(i, out, r0, r1):
            e = torch.cuda.Event(); e.record()
            with torch.cuda.stream(side):
                side.wait_event(e)
                peers[:, r0:r1].copy_(out[:, r0:r1], non_blocking=True)
        with torch.cuda.stream(side):
            for _ in range(8):
                hog_dst.copy_(hog_src, non_blocking=True)
        o, l = att.call_windowed(q, k, v, w, hook, return_softmax_lse=True, static_sched=sched)
        torch.cuda.synchronize()
        assert torch.equal(o, ref_o) and torch.equal(l, ref_l) and torch.equal(att._skip_list, ref_lists), (n, sched)
        assert torch.equal(peers, ref_o)                               # every row was final when its window's hook saw it
