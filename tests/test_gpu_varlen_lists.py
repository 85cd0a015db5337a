"""GPU: skip lists together with ``cu_seqlens`` in ONE launch (round 3; C-ABI: ``read_list`` + ``cu_seqlens_q/k``). The reference's
varlen entry point has no lists (hopper/_internal/flash_attn_interface.py:638-682) and its block-sparse adapter
(flash_attn/flash_blocksparse_attn_interface.py:185-200) binds a kernel that exists nowhere in its csrc, so the checkers are this
library's own fixed-length path (bit-exact: the same kernel walks the same list on the same rows) and the CPU oracle per sequence.
Row (b, h, m) of the lists describes q-tile m of SEQUENCE b over that sequence's own k-tiles; the geometry is that of the maxima."""
import math

import pytest
import torch

from helpers import structured_qkv
from test_gpu_parity import _compare_lists

pytestmark = pytest.mark.gpu


def _pack(parts):
    return torch.cat(parts, dim=0), [0] + torch.tensor([p.shape[0] for p in parts]).cumsum(0).tolist()


@pytest.mark.parametrize("D,dtype", [(128, torch.bfloat16), (64, torch.bfloat16), (80, torch.bfloat16), (96, torch.float16)])
@pytest.mark.parametrize("static", [False, True])
def test_lists_with_cu_seqlens_equal_the_fixed_length_path(D, dtype, static, monkeypatch):
    """Three steps at thr = -2.5 on a packed batch (one empty sequence, one shorter than a tile, ragged lengths, Sq != Sk): every
    step the packed launch must equal, per sequence, a fixed-length launch on that sequence with the same read list - O, LSE and the
    write list bit for bit - and the oracle within the usual bounds. dynamic (ticket queues skip the q-tiles past a sequence's end)
    and static work distribution."""
    import liteattention_amd as L
    from liteattention_amd.flash_attn_interface import mha_fwd
    from oracle import oracle as orc
    if static:
        monkeypatch.setenv("LA_SCHED", "static")
    bm, bn = L.get_tile_sizes(D, 2)
    H, thr = 3, -2.5
    lens_q = [700, 0, 40, 1300, 513]
    lens_k = [900, 64, 130, 1300, 200]
    B = len(lens_q)
    qs, ks, vs = [], [], []
    for b in range(B):
        q, _, _ = structured_qkv(1, max(lens_q[b], 1), H, D, seed=50 + b, alpha=7.0, dtype=torch.float32)
        _, k, v = structured_qkv(1, max(lens_k[b], 1), H, D, seed=50 + b, alpha=7.0, dtype=torch.float32)
        qs.append(q[0, : lens_q[b]].to(dtype)); ks.append(k[0, : lens_k[b]].to(dtype)); vs.append(v[0, : lens_k[b]].to(dtype))
    (qp, cq), (kp, ck), (vp, _) = _pack(qs), _pack(ks), _pack(vs)
    cq_d, ck_d = torch.tensor(cq, dtype=torch.int32).cuda(), torch.tensor(ck, dtype=torch.int32).cuda()
    Qt, Kt = math.ceil(max(lens_q) / bm), math.ceil(max(lens_k) / bn)
    lists = torch.zeros(2, B, H, Qt, Kt + 1, dtype=torch.int32)
    for b in range(B):                                     # initial rows: every tile of the sequence's own key range
        lists[:, b, :, :, 0] = 2
        lists[:, b, :, :, 1] = max(math.ceil(lens_k[b] / bn) - 1, 0)
    lists = lists.cuda()
    must_do = torch.tensor([2, 0, 0], dtype=torch.int32).cuda()
    p_round = "f16" if dtype == torch.float16 else True
    rd = 0
    dropped = 0
    for step in range(3):
        lists[1 - rd].fill_(-7)
        out = torch.full((sum(lens_q), H, D), float("nan"), dtype=dtype, device="cuda")
        o, lse, *_ = mha_fwd(qp.cuda(), kp.cuda(), vp.cuda(), out=out, cu_seqlens_q=cq_d, cu_seqlens_k=ck_d, max_seqlen_q=max(lens_q),
                             max_seqlen_k=max(lens_k), attn_read_list=lists[rd], attn_must_do_list=must_do,
                             attn_write_list=lists[1 - rd], thr=thr, _must_do_is_1d=True)
        assert bool(torch.isfinite(o.float()).all())
        for b in range(B):
            if lens_q[b] == 0:
                continue
            sq, sk = slice(cq[b], cq[b + 1]), slice(ck[b], ck[b + 1])
            qt_b, kt_b = math.ceil(lens_q[b] / bm), math.ceil(lens_k[b] / bn)
            rd_b = lists[rd, b: b + 1, :, :qt_b, : kt_b + 1].contiguous()
            wr_b = torch.zeros_like(rd_b)
            o_b, lse_b, *_ = mha_fwd(qs[b][None].cuda(), ks[b][None].cuda(), vs[b][None].cuda(), attn_read_list=rd_b,
                                     attn_must_do_list=must_do, attn_write_list=wr_b, thr=thr, _must_do_is_1d=True)
            assert torch.equal(o[sq], o_b[0]) and torch.equal(lse[:, sq], lse_b[0]), (step, b)
            got = lists[1 - rd, b, :, :qt_b, : kt_b + 1]
            n = int(wr_b[..., 0].max())
            live = torch.arange(kt_b + 1, device="cuda") <= wr_b[0, ..., 0:1]
            assert bool(((got == wr_b[0]) | ~live).all()), (step, b)
            # and the oracle on that sequence
            margins = torch.empty(1, H, qt_b, kt_b)
            wr_orc = torch.zeros_like(rd_b.cpu())
            o_ref, lse_ref, _ = orc.qkskip_fwd(qs[b][None], ks[b][None], vs[b][None], block_m=bm, block_n=bn, read_list=rd_b.cpu(),
                                               write_list=wr_orc, must_do_list=must_do.cpu(), thr=thr, margins=margins, p_round=p_round)
            ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
            assert (o[sq].float().cpu() - o_ref[0]).abs().max().item() <= ulp * o_ref.abs().max().item() + 1e-3
            assert (lse[:, sq].cpu() - lse_ref[0]).abs().max().item() <= 1e-3
            bad, _ = _compare_lists(orc, rd_b.cpu(), wr_b.cpu(), wr_orc, margins, thr, 1)
            assert bad == 0
            dropped += orc.listed_tiles(rd_b.cpu()) - orc.listed_tiles(wr_b.cpu())
        # rows of q-tiles past a sequence's end and of the empty sequence are not written
        assert bool((lists[1 - rd, 1] == -7).all()) and bool((lists[1 - rd, 2, :, 1:] == -7).all())
        rd = 1 - rd
    assert dropped > 0


def test_block_sparse_packed_batch_is_one_launch_and_matches_the_per_sequence_form():
    """``flash_blocksparse_attn_qkvpacked_func`` (reference signature) on 4 packed sequences: equal, bit for bit, to
    ``flash_blocksparse_attn_func`` called per sequence with the mask's top-left corner; a mask that leaves a q-tile of some
    sequence without k-tiles raises."""
    import liteattention_amd as L
    bm, bn = L.get_tile_sizes(128, 2)
    H, lens = 2, [700, 1100, 64, 333]
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(sum(lens), 3, H, 128, generator=g).bfloat16().cuda()
    cu = [0] + torch.tensor(lens).cumsum(0).tolist()
    max_s = max(lens)
    mask = torch.rand(math.ceil(max_s / bm), math.ceil(max_s / bn), generator=g) < 0.5
    mask[:, 0] = True
    ctx, lse, _ = L.flash_blocksparse_attn_qkvpacked_func(qkv, torch.tensor(cu, dtype=torch.int32).cuda(), mask, 0.0, max_s,
                                                         return_attn_probs=True)
    for b, n in enumerate(lens):
        sl = slice(cu[b], cu[b + 1])
        sub = mask[: math.ceil(n / bm), : math.ceil(n / bn)]
        o_b, lse_b = L.flash_blocksparse_attn_func(qkv[sl, 0][None], qkv[sl, 1][None], qkv[sl, 2][None], sub, return_softmax_lse=True)
        assert torch.equal(ctx[sl], o_b[0]) and torch.equal(lse[:, sl], lse_b[0]), b
    # head_dim 256: lists + cu_seqlens are not built (typed error from the library); the adapter falls back to one launch per sequence
    from liteattention_amd.flash_attn_interface import mha_fwd
    q256 = torch.randn(64, 1, 256, generator=g).bfloat16().cuda()
    c1 = torch.tensor([0, 64], dtype=torch.int32).cuda()
    l256 = L.LiteAttention.init_skip_list(1, 64, 1, 256, False, torch.bfloat16, "cuda")
    with pytest.raises(NotImplementedError):
        mha_fwd(q256, q256, q256, cu_seqlens_q=c1, cu_seqlens_k=c1, max_seqlen_q=64, max_seqlen_k=64, attn_read_list=l256[0],
                attn_write_list=l256[1], thr=-1.0)
    qkv256 = torch.randn(sum(lens[:2]), 3, 1, 256, generator=g).bfloat16().cuda()
    bm2, bn2 = L.get_tile_sizes(256, 2)
    mask2 = torch.rand(math.ceil(1100 / bm2), math.ceil(1100 / bn2), generator=g) < 0.5
    mask2[:, 0] = True
    ctx2 = L.flash_blocksparse_attn_qkvpacked_func(qkv256, torch.tensor(cu[:3], dtype=torch.int32).cuda(), mask2, 0.0, 1100)
    sub = mask2[: math.ceil(700 / bm2), : math.ceil(700 / bn2)]
    assert torch.equal(ctx2[:700], L.flash_blocksparse_attn_func(qkv256[:700, 0][None], qkv256[:700, 1][None], qkv256[:700, 2][None], sub)[0])
    bad = mask.clone()
    bad[1, :] = False
    bad[1, 12] = True                                          # q-tile 1 keeps only k-tile 12: beyond the 333-token sequence's 6 tiles
    with pytest.raises(ValueError):
        L.flash_blocksparse_attn_qkvpacked_func(qkv, torch.tensor(cu, dtype=torch.int32).cuda(), bad, 0.0, max_s)


@pytest.mark.parametrize("D", [128, 64, 96, 192, 256])             # round 6: native fp8 bodies at 64 (prepared V^T tiles of 4 KiB) and 192 / 256 (q-tile 128)
@pytest.mark.parametrize("with_lists", [False, True])
@pytest.mark.parametrize("p_mode", ["encoded", "exp", "reference"])
def test_fp8_packed_batch_equals_the_fixed_length_path(with_lists, p_mode, D, monkeypatch):
    """fp8 (e4m3) with cu_seqlens (round 3): the V^T prepare pass reads cu_seqlens_k itself and writes every sequence's tiles onto the
    [B, Hk, Kt_max] grid, the forward kernel takes each item's rows / lengths from cu_seqlens - ONE forward launch. Per sequence the
    result must equal a fixed-length fp8 launch on that sequence with that sequence's descales - O, LSE and (with lists, three steps at
    thr = -2.5) the write list bit for bit; GQA, an empty query sequence, a sequence without keys, one shorter than a tile."""
    import liteattention_amd as L
    from liteattention_amd.flash_attn_interface import mha_fwd
    monkeypatch.delenv("LA_FP8_P", raising=False)
    if p_mode != "reference":
        monkeypatch.setenv("LA_FP8_P", "mfma_rowsum" if p_mode == "exp" else "encoded")
    F8 = torch.float8_e4m3fn
    H, Hk, thr = 4, 2, -2.5
    bm, bn = L.get_tile_sizes(D, 1)
    lens_q = [700, 0, 40, 1300, 513, 100]
    lens_k = [900, 64, 130, 1300, 200, 0]
    B = len(lens_q)
    qs, ks, vs = [], [], []
    for b in range(B):
        q, _, _ = structured_qkv(1, max(lens_q[b], 1), H, D, seed=70 + b, alpha=7.0, dtype=torch.float32)
        _, k, v = structured_qkv(1, max(lens_k[b], 1), Hk, D, seed=70 + b, alpha=7.0, dtype=torch.float32)
        qs.append(q[0, : lens_q[b]].to(F8)); ks.append(k[0, : lens_k[b]].to(F8)); vs.append(v[0, : lens_k[b]].to(F8))
    (qp, cq), (kp, ck), (vp, _) = _pack(qs), _pack(ks), _pack(vs)
    cq_d, ck_d = torch.tensor(cq, dtype=torch.int32).cuda(), torch.tensor(ck, dtype=torch.int32).cuda()
    g = torch.Generator().manual_seed(5)
    qd, kd, vd = [(0.5 + torch.rand(B, Hk, generator=g)).cuda() for _ in range(3)]
    Qt, Kt = math.ceil(max(lens_q) / bm), math.ceil(max(lens_k) / bn)
    lists = torch.zeros(2, B, H, Qt, Kt + 1, dtype=torch.int32)
    for b in range(B):
        lists[:, b, :, :, 0] = 2
        lists[:, b, :, :, 1] = max(math.ceil(lens_k[b] / bn) - 1, 0)
    lists = lists.cuda()
    must_do = torch.tensor([2, 0, 0], dtype=torch.int32).cuda()
    rd, dropped = 0, 0
    for step in range(3 if with_lists else 1):
        kw = {}
        if with_lists:
            lists[1 - rd].fill_(-7)
            kw = dict(attn_read_list=lists[rd], attn_must_do_list=must_do, attn_write_list=lists[1 - rd], thr=thr, _must_do_is_1d=True)
        out = torch.full((sum(lens_q), H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        o, lse, *_ = mha_fwd(qp.cuda(), kp.cuda(), vp.cuda(), out=out, cu_seqlens_q=cq_d, cu_seqlens_k=ck_d, max_seqlen_q=max(lens_q),
                             max_seqlen_k=max(lens_k), q_descale=qd, k_descale=kd, v_descale=vd, **kw)
        assert o.dtype == torch.bfloat16 and tuple(lse.shape) == (H, sum(lens_q))
        assert bool(torch.isfinite(o.float()).all())
        for b in range(B):
            if lens_q[b] == 0:
                continue
            sq = slice(cq[b], cq[b + 1])
            qt_b, kt_b = math.ceil(lens_q[b] / bm), math.ceil(lens_k[b] / bn)
            kw_b = {}
            if with_lists and lens_k[b] > 0:
                rd_b = lists[rd, b: b + 1, :, :qt_b, : kt_b + 1].contiguous()
                wr_b = torch.zeros_like(rd_b)
                kw_b = dict(attn_read_list=rd_b, attn_must_do_list=must_do, attn_write_list=wr_b, thr=thr, _must_do_is_1d=True)
            o_b, lse_b, *_ = mha_fwd(qs[b][None].cuda(), ks[b][None].cuda(), vs[b][None].cuda(), q_descale=qd[b: b + 1],
                                     k_descale=kd[b: b + 1], v_descale=vd[b: b + 1], **kw_b)
            assert torch.equal(o[sq], o_b[0]) and torch.equal(lse[:, sq], lse_b[0]), (step, b)
            if lens_k[b] == 0:
                assert bool((o[sq] == 0).all()) and bool(torch.isinf(lse[:, sq]).all())          # flash_api.cpp:1241-1245
            elif with_lists:
                got = lists[1 - rd, b, :, :qt_b, : kt_b + 1]
                live = torch.arange(kt_b + 1, device="cuda") <= wr_b[0, ..., 0:1]
                assert bool(((got == wr_b[0]) | ~live).all()), (step, b)
                dropped += int((rd_b[..., 0] != wr_b[..., 0]).sum()) + int((rd_b[..., 1:3] != wr_b[..., 1:3]).sum())
        rd = 1 - rd
    assert dropped > 0 or not with_lists
    with pytest.raises(RuntimeError, match="descale"):
        mha_fwd(qp.cuda(), kp.cuda(), vp.cuda(), cu_seqlens_q=cq_d, cu_seqlens_k=ck_d, max_seqlen_q=max(lens_q),
                max_seqlen_k=max(lens_k), q_descale=qd[:2])
