"""GPU parity tests (run with -m gpu on an MI355X). Everything goes through the C-ABI (la_fwd) of the
HIP library; the CPU oracle (oracle/) and the committed reference outputs (tests/golden) are the checkers.

Tolerances (stated once):
  * dense O vs the reference's fp32 `out_ref`: the reference's own rule, hopper/tests/test_flash_attn.py:296
        max|out - out_ref| <= 2 * max|out_pt - out_ref| + 2 * max|out_ref + 0.3 - 0.3 - out_ref|
  * dense O vs the tiled oracle at the same tile sizes (same algorithm, fp32 accumulate, bf16 P):
        max|out - oracle| <= 2^-8 * max|oracle| + 1e-3        (one bf16 output rounding + summation order)
  * LSE (fp32): max|lse - ref| <= 1e-3 (the reference script accepts 0.1, test_lite_attention.py:90)
  * skip lists: bit-exact, except rows holding a tile whose decision margin |max_r (m_loc-m_prev)c - thr|
    is below 1e-3 (fp32 summation order can flip such a tile; SURVEY.md §7 hard part 5).
"""
import ctypes
import math

import pytest
import torch

from helpers import DENSE_CASES, load_dense_case, ref_tolerance, structured_qkv

pytestmark = pytest.mark.gpu

def _tiles(head_dim):
    """The skip lists follow the tile of the kernel the library selected (LA_FWD_KERNEL): ask it, as LiteAttention.get_MN does."""
    import liteattention_amd as L
    return L.get_tile_sizes(head_dim, 2)


BM, BN = _tiles(128)          # (128, 64) for the default kernels, (256, 64) for the 64-rows-per-wave kernel
BM64, BN64 = _tiles(64)


def _L():
    import liteattention_amd as L
    assert L.get_tile_sizes(128, 2) in ((128, 64), (256, 64))
    return L


def _orc():
    from oracle import oracle as orc
    return orc


def _randn(B, S, H, D=128, seed=0, Sk=None):  # noqa: E302
    g = torch.Generator().manual_seed(seed)
    Sk = S if Sk is None else Sk
    q = torch.randn(B, S, H, D, generator=g).bfloat16()
    k = torch.randn(B, Sk, H, D, generator=g).bfloat16()
    v = torch.randn(B, Sk, H, D, generator=g).bfloat16()
    return q, k, v


def _oracle_tol(o_ref):
    return 2.0 ** -8 * o_ref.abs().max().item() + 1e-3


# ------------------------------------------------------------------------------------------ dense
@pytest.mark.parametrize("name", [n for n in DENSE_CASES if "fp32" not in n])      # the fp32 case (configs[0]): tests/test_gpu_round4.py
def test_dense_matches_reference_outputs(name):
    L = _L()
    c = load_dense_case(name)
    q, k, v = [x.to(torch.bfloat16).cuda() for x in (c["q"], c["k"], c["v"])]
    out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    assert out.dtype == torch.bfloat16 and out.shape == q.shape and lse.shape == (q.shape[0], q.shape[2], q.shape[1])
    err = (out.float().cpu() - c["out_ref"]).abs().max().item()
    assert err <= ref_tolerance(c["out_ref"], c["pt_maxerr"]), (err, ref_tolerance(c["out_ref"], c["pt_maxerr"]))
    assert (lse.cpu() - c["lse_ref"]).abs().max().item() <= 1e-3


@pytest.mark.parametrize("shape", [(1, 17, 1, 17), (1, 64, 2, 64), (2, 129, 3, 65), (1, 1000, 2, 1000),
                                   (3, 257, 5, 640), (1, 128, 1, 4224), (1, 1023, 1, 1024)])
def test_dense_matches_tiled_oracle_ragged_shapes(shape):
    """Empty-ish, ragged and odd sizes (the reference grid: test_flash_attn.py:152-177)."""
    L, orc = _L(), _orc()
    B, Sq, H, Sk = shape
    q, k, v = _randn(B, Sq, H, seed=Sq + Sk, Sk=Sk)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN)
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    # the dense eager oracle agrees as well (reference rule)
    o_eager, _ = orc.attention_dense_ref(q, k, v)
    o_pt, _ = orc.attention_dense_ref(q, k, v, upcast=False, reorder_ops=True)
    tol = orc.dense_tolerance(o_eager.float(), o_pt)
    assert (out.float().cpu() - o_eager.float()).abs().max().item() <= tol + 2.0 ** -8 * o_eager.float().abs().max().item()


@pytest.mark.parametrize("gain", [3.0, 8.0])
def test_running_max_that_grows_late_in_the_walk(gain):
    """Keys are walked from the END of the sequence; make the early keys (walked last) score far higher, so the running
    max of every row keeps growing by many powers of two through the walk. Exercises the O / row-sum rescale — in the
    64-rows-per-wave kernel the lazy-rescale path (m_ref follows m_true only past 2^8) and its O^T AGPR round trip — with
    and without skip lists; the lists must still match the oracle bit for bit (votes use the true running max)."""
    L, orc = _L(), _orc()
    B, S, H = 1, 1536, 2
    q, k, v = _randn(B, S, H, seed=91)
    ramp = torch.linspace(gain, 1.0, S).view(1, S, 1, 1)                 # early keys (walked last) up to `gain` x larger
    k = (k.float() * ramp).bfloat16()
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN)
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    # same data through the skip path (threshold low enough that nothing is skipped: every tile raises the max)
    Qt, Kt = -(-S // BM), -(-S // BN)
    att = L.LiteAttention(threshold=-1.0, max_batch_size=B)
    md_row = orc.expand_must_do_ref([0, 0], BN, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    for _ in range(2):
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc,
                                           must_do_list=md_row, thr=-1.0, margins=margins)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, -1.0, B)
        assert bad == 0


def test_softmax_scale_and_strided_inputs():
    """Non-default scale; q/k/v as non-contiguous views of a packed (B,S,3,H,D) tensor."""
    L, orc = _L(), _orc()
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(2, 300, 3, 4, 128, generator=g).bfloat16()
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    assert not q.is_contiguous()
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, softmax_scale=0.05)
    qkv_d = qkv.cuda()
    out, lse = L.flash_attn_func(qkv_d[:, :, 0], qkv_d[:, :, 1], qkv_d[:, :, 2], softmax_scale=0.05,
                                 return_softmax_lse=True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3


def test_empty_key_sequence():
    L = _L()
    q = torch.randn(1, 5, 2, 128).bfloat16().cuda()
    k = torch.zeros(1, 0, 2, 128, dtype=torch.bfloat16, device="cuda")
    out, lse = L.flash_attn_func(q, k, k, return_softmax_lse=True)
    assert (out == 0).all() and torch.isinf(lse).all() and (lse > 0).all()      # flash_api.cpp:1241-1245


def test_rescale_branch_is_exercised():
    """A key spike late in the walk forces a large running-max jump (guide rule 26)."""
    L, orc = _L(), _orc()
    q, k, v = _randn(1, 512, 1, seed=3)
    k = k.float()
    k[0, 100] = q[0, 300, 0].float() * 6.0          # tile 1 (processed late): huge score for query 300
    k = k.bfloat16()
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN)
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    assert torch.isfinite(out.float()).all()


# ------------------------------------------------------------------------------ known answers K1-K4
def test_reference_script_known_answers():
    """/root/reference/test_lite_attention.py:7-93 at its own shape (2,5000,32,128) bf16, seed 0."""
    L = _L()
    torch.manual_seed(0)
    q, k, v = [torch.randn(2, 5000, 32, 128, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    Kt = math.ceil(5000 / BN)
    # skip all
    attn = L.LiteAttention()
    attn.threshold = float("inf")
    attn(q, k, v)
    assert (attn._skip_list[1, ..., 0] <= 2).all()
    assert (attn._skip_list[1, :2, ..., 1] == Kt - 1).all() and (attn._skip_list[1, :2, ..., 2] == Kt - 2).all()
    # must do
    attn = L.LiteAttention()
    attn.threshold = float("inf")
    attn(q, k, v, must_do_list=[k.shape[1] - 1, 0])
    assert (attn._skip_list[1] == attn._skip_list[0]).all()
    # skip nothing
    attn = L.LiteAttention()
    attn.threshold = float("-inf")
    attn(q, k, v)
    assert (attn._skip_list[1] == attn._skip_list[0]).all()
    # LSE
    attn = L.LiteAttention()
    attn.threshold = 0.0
    out, lse = attn(q, k, v, return_softmax_lse=True)
    scale = 1.0 / 128 ** 0.5
    worst = 0.0
    for b in range(2):
        for h0 in range(0, 32, 8):
            qr = q[b, :, h0:h0 + 8].transpose(0, 1).float()
            kr = k[b, :, h0:h0 + 8].transpose(0, 1).float()
            lse_ref = torch.logsumexp(torch.matmul(qr, kr.transpose(-2, -1)) * scale, dim=-1)
            worst = max(worst, (lse_ref - lse[b, h0:h0 + 8]).abs().max().item())
    assert worst < 1e-3         # reference accepts < 0.1


# --------------------------------------------------------------------------- temporal list parity
def _compare_lists(orc, rd_cpu, wr_gpu_cpu, wr_orc, margins, thr, B):
    """Bit-exact except rows with a borderline tile."""
    bad_rows = 0
    borderline_rows = 0
    H, Qt = wr_orc.shape[1], wr_orc.shape[2]
    for b in range(B):
        for h in range(H):
            for m in range(Qt):
                a, e = wr_gpu_cpu[b, h, m], wr_orc[b, h, m]
                L0 = int(e[0])
                if int(a[0]) == L0 and torch.equal(a[: L0 + 1], e[: L0 + 1]):
                    continue
                mg = margins[b, h, m]
                mg = mg[~torch.isnan(mg)]
                if ((mg - thr).abs() < 1e-3).any():
                    borderline_rows += 1
                else:
                    bad_rows += 1
    return bad_rows, borderline_rows


@pytest.mark.parametrize("thr", [-2.0, -6.0])
@pytest.mark.parametrize("use_must_do", [False, True, "two_ranges"])
def test_multi_step_lists_match_oracle(thr, use_must_do):
    """5 denoising-like steps with slowly varying structured inputs. At every step the oracle gets the
    SAME read list as the kernel; write lists must be identical, outputs within the oracle tolerance."""
    L, orc = _L(), _orc()
    B, S, H = 2, 1536, 2
    Qt, Kt = S // BM, S // BN
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    # one range -> wave-parallel writer; two ranges -> the literal (stateful) must-do reader path
    must_do = [1300, 1100, 700, 400] if use_must_do == "two_ranges" else ([700, 400] if use_must_do else None)
    md_row = orc.expand_must_do_ref(must_do if use_must_do else [0, 0], BN, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    total_border = 0
    listed = []
    for step in range(5):
        q, k, v = structured_qkv(B, S, H, 128, seed=100)      # same base...
        g = torch.Generator().manual_seed(1000 + step)          # ...plus a small per-step perturbation
        q = (q.float() + 0.05 * torch.randn(q.shape, generator=g)).bfloat16()
        k = (k.float() + 0.05 * torch.randn(k.shape, generator=g)).bfloat16()
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, must_do_list=must_do)
        rd = att._skip_list[rd_idx].cpu()
        wr = att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc,
                                                 must_do_list=md_row, thr=thr, margins=margins)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
        bad, border = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0, f"step {step}: {bad} rows differ from the oracle with no borderline tile"
        total_border += border
        listed.append(orc.listed_tiles(wr[:B]))
        assert n_tiles == orc.listed_tiles(rd[:B])
    assert listed == sorted(listed, reverse=True)                # sparsity is monotone
    assert listed[-1] < 0.9 * B * H * Qt * Kt                    # and real
    assert total_border <= 2


def test_sparse_run_with_ragged_tail_and_partial_q_tile():
    """Sq, Sk not multiples of the tiles: zero query rows of the last q-tile take part in the vote
    (TMA zero fill in the reference) and the first walked tile carries the seqlen mask."""
    L, orc = _L(), _orc()
    B, Sq, Sk, H = 1, 1100, 1250, 2
    Qt, Kt = math.ceil(Sq / BM), math.ceil(Sk / BN)
    q, _, _ = structured_qkv(B, Sq, H, 128, seed=7)
    _, k, v = structured_qkv(B, Sk, H, 128, seed=7)
    thr = -3.0
    att = L.LiteAttention(threshold=thr, max_batch_size=1)
    margins = torch.empty(B, H, Qt, Kt)
    for step in range(3):
        rd_idx = att._phase if att._skip_list is not None else 0
        out = att(q.cuda(), k.cuda(), v.cuda())
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, _, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc, thr=thr,
                                     margins=margins)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0
    # the padded last q-tile never drops anything at a negative threshold (zero rows give margin 0 > thr)
    last = att._skip_list[att._phase, 0, :, Qt - 1].cpu()
    assert (last[:, 0] == 2).all() and (last[:, 1] == Kt - 1).all() and (last[:, 2] == 0).all()


def test_full_list_equals_dense_kernel_bit_exactly():
    """Walking a list that names every tile (in 1 or 3 ranges) must equal the dense kernel bit for bit."""
    L = _L()
    q, k, v = [x.cuda() for x in _randn(1, 2048, 3, seed=5)]
    dense, lse_d = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    Qt, Kt = 2048 // BM, 2048 // BN
    lists = L.LiteAttention.init_skip_list(1, 2048, 3, 128, False, torch.bfloat16, "cuda")
    lists[0, ..., :7] = torch.tensor([6, Kt - 1, 20, 19, 7, 6, 0], dtype=torch.int32, device="cuda")
    md = torch.tensor([2, 0, 0], dtype=torch.int32, device="cuda")
    out, lse = L.flash_attn_func(q, k, v, attn_read_list=lists[0], attn_write_list=lists[1], attn_must_do_list=md,
                                 thr=float("-inf"), return_softmax_lse=True)
    assert torch.equal(out, dense) and torch.equal(lse, lse_d)
    # thr=-inf keeps every tile: the written list covers the same tiles, merged into one range
    assert (lists[1][..., 0] == 6).all() or (lists[1][..., 0] == 2).all()
    from oracle import oracle as orc
    assert orc.walk_tiles(lists[1][0, 0, 0].cpu().tolist()) == list(range(Kt - 1, -1, -1))


def test_reference_4d_must_do_tensor_through_the_registered_op():
    """The reference passes a [maxB,H,Qt,Kt+1] must-do tensor through torch.ops.lite_attention.fwd."""
    L, orc = _L(), _orc()
    B, S, H = 1, 1024, 2
    q, k, v = structured_qkv(B, S, H, 128, seed=21)
    Qt, Kt = S // BM, S // BN
    lists = L.LiteAttention.init_skip_list(B, S, H, 128, False, torch.bfloat16, "cuda")
    md4 = L.LiteAttention._expand_must_do_list([600, 200], (B, H, Qt, Kt + 1), q.cuda(), v.cuda())
    out = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), attn_read_list=lists[0], attn_must_do_list=md4,
                            attn_write_list=lists[1], thr=-1.0)
    rd = lists[0].cpu()
    wr_orc = torch.zeros_like(rd)
    orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc,
                   must_do_list=md4.cpu(), thr=-1.0)
    wr = lists[1].cpu()
    for h in range(H):
        for m in range(Qt):
            n = int(wr_orc[0, h, m, 0])
            assert wr[0, h, m, : n + 1].tolist() == wr_orc[0, h, m, : n + 1].tolist()


def test_determinism_including_lists():
    """Race screen in the spirit of test_flash_attn.py:1144-1175: repeated launches are bit-identical."""
    L = _L()
    q, k, v = [x.cuda() for x in structured_qkv(2, 2000, 4, 128, seed=33)]
    ref_out, ref_lists = None, None
    for it in range(25):
        att = L.LiteAttention(threshold=-3.0, max_batch_size=2)
        att(q, k, v)
        out = att(q, k, v)
        lists = att._skip_list.clone()
        if ref_out is None:
            ref_out, ref_lists = out.clone(), lists
        assert torch.equal(out, ref_out) and torch.equal(lists, ref_lists), f"iteration {it}"


# ------------------------------------------------------------------------------ errors (loud, typed)
def test_error_behaviour():
    L = _L()
    q = torch.randn(1, 256, 2, 128, device="cuda").bfloat16()
    assert L.flash_attn_func(q.half(), q.half(), q.half()).dtype == torch.float16      # built since round 2 (tests/test_gpu_fp16.py)
    with pytest.raises(RuntimeError):
        L.flash_attn_func(q.float(), q.float(), q.float())
    with pytest.raises(NotImplementedError):
        L.flash_attn_func(q, q, q, causal=True)
    with pytest.raises(NotImplementedError):
        L.flash_attn_func(q, q, q, window_size=(16, 0))
    with pytest.raises(NotImplementedError):
        L.flash_attn_func(q, q, q, softcap=1.0)
    with pytest.raises(RuntimeError, match="must divide"):
        q3 = torch.randn(1, 256, 3, 128, device="cuda").bfloat16()
        L.flash_attn_func(q3, q3[:, :, :2], q3[:, :, :2])                      # nheads_k must divide nheads (:777)
    with pytest.raises(NotImplementedError):
        L.flash_attn_func(q, q, q[..., :64])                                   # head_dim_v != head_dim
    q320 = torch.randn(1, 256, 2, 320, device="cuda").bfloat16()
    with pytest.raises(RuntimeError, match="head_size"):
        L.flash_attn_func(q320, q320, q320)                                    # head_dim > 256 not instantiated
    lists = torch.zeros(1, 2, 256 // BM, 5, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="attn_read_list"):
        L.flash_attn_func(q, q, q, attn_read_list=lists.long(), attn_write_list=lists)
    with pytest.raises(RuntimeError, match="shape"):
        L.flash_attn_func(q, q, q, attn_read_list=lists[..., :4].contiguous(), attn_write_list=lists[..., :4].contiguous())
    with pytest.raises(RuntimeError, match="together"):
        L.flash_attn_func(q, q, q, attn_read_list=lists)
    x = torch.randn(1, 256, 2, 128, device="cuda").bfloat16().requires_grad_()
    out = L.flash_attn_func(x, x, x)
    with pytest.raises(NotImplementedError):
        out.sum().backward()


# ----------------------------------------------------------------------- raw C-ABI (no torch op)
def test_raw_cabi_call_with_plain_pointers():
    from liteattention_amd import _cabi
    orc = _orc()
    q, k, v = _randn(1, 384, 2, seed=8)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    o = torch.empty_like(qd)
    lse = torch.empty(1, 2, 384, dtype=torch.float32, device="cuda")
    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(a)
    a.dtype = _cabi.LA_DTYPE_BF16
    a.q, a.k, a.v, a.o, a.lse = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), lse.data_ptr()
    for name in "qkvo":
        setattr(a, f"{name}_batch_stride", 384 * 2 * 128)
        setattr(a, f"{name}_row_stride", 2 * 128)
        setattr(a, f"{name}_head_stride", 128)
    a.batch, a.seqlen_q, a.seqlen_k, a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = 1, 384, 384, 2, 2, 128, 128
    a.softmax_scale = 128 ** -0.5
    a.block_m, a.block_n = BM, BN
    a.flags = _cabi.default_flags() & _cabi.GEOMETRY_FLAGS           # (BM, BN) is the tile of the kernel the environment selects
    rc = _cabi.load().la_fwd(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _cabi.status_string(rc)
    torch.cuda.synchronize()
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN)
    assert (o.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3


# ------------------------------------------------------------------------ helpers around the path
def test_skip_list_stats_kernel_and_skip_fraction():
    L, orc = _L(), _orc()
    q, k, v = [x.cuda() for x in structured_qkv(2, 1536, 3, 128, seed=44)]
    att = L.LiteAttention(threshold=-2.0, max_batch_size=4)
    for _ in range(3):
        att(q, k, v)
    rl = att.current_read_list()
    counts = L.skip_list_stats(rl, batch=2).cpu()
    assert counts[0].item() == orc.listed_tiles(rl[:2].cpu()) and counts[1].item() == 2 * 3 * (1536 // BM)
    frac = att.get_skip_fraction(batch=2)
    assert abs((1 - frac) - orc.listed_tiles(rl[:2].cpu()) / (2 * 3 * (1536 // BM) * 24)) < 1e-9 and frac > 0.05
    assert abs(L.LiteAttention.calc_percentage(rl[:2]) - L.LiteAttention.calc_percentage(rl[:2].cpu())) < 1e-12


def test_seq_parallel_splits_plus_combine_equal_full_attention():
    """SeqParallelLiteAttention (one skip state per K/V split) + flash_attn_combine == one full call
    (README.md:199-250 recipe; merge oracle test_flash_attn.py:1178-1187)."""
    L, orc = _L(), _orc()
    q, k, v = _randn(1, 1024, 2, seed=17)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    sp = L.SeqParallelLiteAttention(num_nodes=4, threshold=-20.0, max_batch_size=1)
    outs, lses = [], []
    for j in range(4):
        o, l = sp(qd, kd[:, j * 256:(j + 1) * 256], vd[:, j * 256:(j + 1) * 256], split_idx=j, return_softmax_lse=True)
        outs.append(o)
        lses.append(l)
    assert sp.lite_attention[2]._skip_list.shape == (2, 1, 2, 1024 // BM, 5)  # Kt from the split length
    out, lse = L.flash_attn_combine(torch.stack(outs), torch.stack(lses))
    o_ref, lse_ref = orc.attention_dense_ref(q, k, v)
    o_c, lse_c = orc.attention_combine_ref(torch.stack(outs).float().cpu(), torch.stack(lses).transpose(2, 3).cpu())
    assert (out.float().cpu() - o_c).abs().max().item() <= 2.0 ** -8 * o_c.abs().max().item() + 1e-4
    assert (lse.cpu() - lse_c.transpose(1, 2)).abs().max().item() <= 1e-4
    assert (out.float().cpu() - o_ref.float()).abs().max().item() <= 3e-2
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    # fp32 partials: an fp32 result by default, as in the reference (hopper/_internal/flash_attn_interface.py:684-685: the output
    # dtype defaults to that of the partials) - it is the merge oracle to fp32 round-off; bf16 on request, equal to the bf16 merge
    out32, lse32 = L.flash_attn_combine(torch.stack(outs).float(), torch.stack(lses))
    assert out32.dtype == torch.float32
    assert (out32.cpu() - o_c).abs().max().item() <= 2e-6 and torch.equal(lse32, lse)
    out16, _ = L.flash_attn_combine(torch.stack(outs).float(), torch.stack(lses), out_dtype=torch.bfloat16)
    assert torch.equal(out16, out)
    with pytest.raises(RuntimeError, match="16-bit partial"):
        L.flash_attn_combine(torch.stack(outs), torch.stack(lses), out_dtype=torch.float32)


# -------------------------------------------------------------------- full-size properties (C2 shape)
def test_full_size_properties_c2():
    """BASELINE.json configs[1]: S=32768, H=40, D=128 bf16. The oracle cannot finish this size, so check
    size-independent properties: fp32 torch reference on sampled rows, list fixed point at thr=-inf,
    full-list == dense bit-exactly, linearity in V."""
    L = _L()
    S, H = 32768, 40
    g = torch.Generator(device="cuda").manual_seed(0)
    q, k, v = [torch.randn(1, S, H, 128, device="cuda", generator=g, dtype=torch.float32).bfloat16() for _ in range(3)]
    att = L.LiteAttention(max_batch_size=1)
    att.threshold = float("-inf")
    out, lse = att(q, k, v, return_softmax_lse=True)
    assert torch.equal(att._skip_list[0], att._skip_list[1])                 # nothing dropped: fixed point
    dense, lse_d = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    assert torch.equal(out, dense) and torch.equal(lse, lse_d)
    # sampled rows against a plain fp32 torch reference
    rows = torch.randint(0, S, (192,), generator=torch.Generator().manual_seed(1)).cuda()
    for h in (0, 17, 39):
        s = (q[0, rows, h].float() @ k[0, :, h].float().T) * (128 ** -0.5)
        ref_lse = torch.logsumexp(s, dim=-1)
        ref_o = torch.softmax(s, dim=-1) @ v[0, :, h].float()
        assert (lse[0, h, rows] - ref_lse).abs().max().item() <= 1e-3
        assert (out[0, rows, h].float() - ref_o).abs().max().item() <= 2.0 ** -8 * ref_o.abs().max().item() + 2e-3
    # linearity in V: O(v1 + v2) == O(v1) + O(v2) up to bf16 rounding of inputs/outputs
    v2 = torch.randn(1, S, H, 128, device="cuda", generator=g, dtype=torch.float32).bfloat16()
    vsum = (v.float() + v2.float()).bfloat16()
    o2 = L.flash_attn_func(q, k, v2)
    osum = L.flash_attn_func(q, k, vsum)
    assert (osum.float() - (dense.float() + o2.float())).abs().max().item() <= 3e-2


# ----------------------------------------------------------------------------------- head_dim 64
@pytest.mark.parametrize("shape", [(1, 64, 1, 64), (2, 333, 3, 333), (1, 1000, 2, 1250), (1, 2048, 1, 2048)])
def test_head_dim_64_dense_matches_oracle(shape):
    """Second instantiation of the reference's default build (hopper/instantiations/flash_fwd_hdim64_bf16_sm90.cu);
    (1,2048,1,64) is BASELINE.json configs[0]'s shape in bf16."""
    import liteattention_amd as L
    orc = _orc()
    assert L.get_tile_sizes(64, 2) == (BM64, BN64) and BN64 == 64 and BM64 in (256, 128)     # 128 under LA_FWD_KERNEL=v2
    B, Sq, H, Sk = shape
    q, k, v = _randn(B, Sq, H, D=64, seed=Sq, Sk=Sk)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM64, block_n=BN64)
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    o_eager, _ = orc.attention_dense_ref(q, k, v)
    o_pt, _ = orc.attention_dense_ref(q, k, v, upcast=False, reorder_ops=True)
    assert (out.float().cpu() - o_eager.float()).abs().max().item() <= \
        orc.dense_tolerance(o_eager.float(), o_pt) + 2.0 ** -8 * o_eager.float().abs().max().item()


def test_head_dim_64_skip_lists_match_oracle():
    import liteattention_amd as L
    orc = _orc()
    B, S, H, thr = 1, 1536, 3, -3.0
    Qt, Kt = S // BM64, S // BN64
    assert S % BM64 == 0
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    md_row = orc.expand_must_do_ref([0, 0], BN64, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    listed = []
    for step in range(4):
        q, k, v = structured_qkv(B, S, H, 64, seed=200, alpha=7.0)
        g = torch.Generator().manual_seed(3000 + step)
        q = (q.float() + 0.05 * torch.randn(q.shape, generator=g)).bfloat16()
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM64, block_n=BN64, read_list=rd, write_list=wr_orc,
                                           must_do_list=md_row, thr=thr, margins=margins)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _oracle_tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0
        listed.append(orc.listed_tiles(wr[:B]))
    assert listed[-1] < 0.9 * B * H * Qt * Kt
    # reference script checks at head_dim 64 (test_lite_attention.py:7 loops over head dims)
    q, k, v = [x.cuda() for x in _randn(2, 5000, 8, D=64, seed=0)]
    attn = L.LiteAttention()
    attn.threshold = float("inf")
    attn(q, k, v)
    assert (attn._skip_list[1, ..., 0] <= 2).all()
    attn = L.LiteAttention()
    attn.threshold = float("-inf")
    attn(q, k, v)
    assert (attn._skip_list[1] == attn._skip_list[0]).all()
