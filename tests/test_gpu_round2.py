"""GPU: round-2 rows — the operator surface under its own import names (compat_shims), one-launch varlen, ``must_skip_list``
and checkpoint/resume through the kernel, the ABI-4 boundary (seqlen_k == 0 in the library, kernel selection
by flag) and the 1000-iteration race screen of the reference (hopper/tests/test_flash_attn.py:1144-1175).
Checker: the CPU oracle at the kernel's tile sizes; tolerances as in test_gpu_parity.py."""
import ctypes
import importlib.util
import math
import os
import sys

import pytest
import torch

from helpers import structured_qkv, fp8_lse_tol, fp8_p_round

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiles(d=128, es=2):
    import liteattention_amd as L
    return L.get_tile_sizes(d, es)


BM, BN = _tiles()


def _tol(o):
    return 2.0 ** -8 * o.abs().max().item() + 1e-3


def _packed(lens_q, lens_k, H, D, seed, Hk=None):
    g = torch.Generator().manual_seed(seed)
    Hk = H if Hk is None else Hk
    q = torch.randn(sum(lens_q), H, D, generator=g).bfloat16()
    k = torch.randn(sum(lens_k), Hk, D, generator=g).bfloat16()
    v = torch.randn(sum(lens_k), Hk, D, generator=g).bfloat16()
    cq = [0] + torch.tensor(lens_q).cumsum(0).tolist()
    ck = [0] + torch.tensor(lens_k).cumsum(0).tolist()
    return q, k, v, cq, ck


# ------------------------------------------------------------------------------------------ f2: import names + varlen
@pytest.fixture
def shims():
    """compat_shims/ on sys.path, as a user opts in; removed (and the modules forgotten) afterwards."""
    path = os.path.join(ROOT, "compat_shims")
    assert importlib.util.find_spec("flash_attn") is None, "a real flash_attn install is present: the shim test would shadow it"
    sys.path.insert(0, path)
    try:
        yield path
    finally:
        sys.path.remove(path)
        for name in [n for n in sys.modules if n == "flash_attn" or n.startswith("flash_attn.") or n == "flash_attn_interface"]:
            del sys.modules[name]


def test_stock_import_names_resolve_to_the_gfx950_kernel(shims):
    """`from flash_attn import flash_attn_func, flash_attn_varlen_func`, `import flash_attn_interface` (FA3) and
    `flash_attn.flash_blocksparse_attn_interface` — what a stock Wan2.x pipeline imports — run on the HIP kernel and match the
    oracle (reference modules: flash_attn/flash_attn_interface.py:1135,1370; hopper/_internal/flash_attn_interface.py:547,638;
    flash_attn/flash_blocksparse_attn_interface.py:185-200)."""
    from oracle import oracle as orc
    import flash_attn
    import flash_attn_interface as fa3
    from flash_attn.flash_blocksparse_attn_interface import flash_blocksparse_attn_func
    assert flash_attn.__file__.startswith(shims) and fa3.__file__.startswith(shims)
    g = torch.Generator().manual_seed(8)
    q, k, v = [torch.randn(2, 300, 4, 128, generator=g).bfloat16() for _ in range(3)]
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN)
    # FA2 dense (+ the (out, lse, None) convention), kv-packed and qkv-packed views
    out, lse, s_dmask = flash_attn.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_attn_probs=True)
    assert s_dmask is None and (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    qkv = torch.stack([q, k, v], dim=2).cuda()
    assert torch.equal(flash_attn.flash_attn_qkvpacked_func(qkv), out)
    assert torch.equal(flash_attn.flash_attn_kvpacked_func(q.cuda(), qkv[:, :, 1:]), out)
    # FA3 names: dense func, and the LiteAttention extension arguments are there
    assert torch.equal(fa3.flash_attn_func(q.cuda(), k.cuda(), v.cuda()), out)
    import inspect
    assert "attn_read_list" in inspect.signature(fa3.flash_attn_func).parameters
    # varlen through both dialects: ONE launch on packed tensors with device cu_seqlens
    lens_q, lens_k = [300, 77, 512, 0, 40], [512, 200, 64, 30, 0]        # cross-attention-like; an empty q and an empty k sequence
    qp, kp, vp, cq, ck = _packed(lens_q, lens_k, 4, 128, seed=3)
    cq_d, ck_d = torch.tensor(cq, dtype=torch.int32).cuda(), torch.tensor(ck, dtype=torch.int32).cuda()
    o2, lse2, _ = flash_attn.flash_attn_varlen_func(qp.cuda(), kp.cuda(), vp.cuda(), cq_d, ck_d, max(lens_q), max(lens_k),
                                                    return_attn_probs=True)
    o3 = fa3.flash_attn_varlen_func(qp.cuda(), kp.cuda(), vp.cuda(), cq_d, ck_d, max(lens_q), max(lens_k))
    assert o2.dtype == torch.bfloat16 and lse2.shape == (4, sum(lens_q)) and torch.equal(o2, o3)
    for b in range(len(lens_q)):
        if lens_q[b] == 0:
            continue
        sl_q, sl_k = slice(cq[b], cq[b + 1]), slice(ck[b], ck[b + 1])
        if lens_k[b] == 0:                                                 # no keys: o = 0, lse = +inf (flash_api.cpp:1241-1245)
            assert (o2[sl_q] == 0).all() and torch.isinf(lse2[:, sl_q]).all() and (lse2[:, sl_q] > 0).all()
            continue
        o_r, lse_r, _ = orc.qkskip_fwd(qp[sl_q][None], kp[sl_k][None], vp[sl_k][None], block_m=BM, block_n=BN)
        assert (o2[sl_q].float().cpu() - o_r[0]).abs().max().item() <= _tol(o_r)
        assert (lse2[:, sl_q].cpu() - lse_r[0]).abs().max().item() <= 1e-3
    # block-sparse, reference signature: packed qkv + cu_seqlens + one block mask over this kernel's tiles
    lens = [700, 1100]
    qb, kb, vb, cu, _ = _packed(lens, lens, 2, 128, seed=5)
    qkv_p = torch.stack([qb, kb, vb], dim=1).cuda()
    max_s = max(lens)
    mask = torch.rand(math.ceil(max_s / BM), math.ceil(max_s / BN), generator=g) < 0.5
    mask[:, 0] = True
    ctx, lse_b, _ = flash_blocksparse_attn_func(qkv_p, torch.tensor(cu, dtype=torch.int32).cuda(), mask, 0.0, max_s,
                                                return_attn_probs=True)
    for b in range(2):
        sl = slice(cu[b], cu[b + 1])
        qt, kt = math.ceil(lens[b] / BM), math.ceil(lens[b] / BN)
        keep = mask[:qt, :kt].repeat_interleave(BM, 0)[: lens[b]].repeat_interleave(BN, 1)[:, : lens[b]]
        s = torch.einsum("qhd,khd->hqk", qb[sl].float(), kb[sl].float()) * 128 ** -0.5
        s = s.masked_fill(~keep[None], float("-inf"))
        ref = torch.einsum("hqk,khd->qhd", torch.softmax(s, -1), vb[sl].float())
        assert (ctx[sl].float().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 2e-3
        assert (lse_b[:, sl].cpu() - torch.logsumexp(s, -1)).abs().max().item() <= 1e-3


@pytest.mark.parametrize("D,H,Hk", [(128, 4, 4), (64, 4, 2), (96, 2, 2), (192, 4, 2), (256, 2, 1), (80, 2, 2)])
def test_varlen_is_one_launch_for_every_bf16_instantiation(D, H, Hk):
    """Packed batches on every bf16 kernel (head_dim 64 / 128 / 256, 96 zero-padded onto 128), MHA and GQA/MQA, ragged lengths
    around the tile sizes; nothing but the listed rows is written (NaN-prefilled out stays NaN past total_q... there is no
    such row: instead rows of EMPTY q sequences do not exist and every real row must be finite)."""
    import liteattention_amd as L
    from oracle import oracle as orc
    bm, bn = _tiles(D)
    lens_q = [1, 255, 256, 257, 600, 64]
    lens_k = [64, 65, 1, 130, 999, 640]
    q, k, v, cq, ck = _packed(lens_q, lens_k, H, D, seed=D + H, Hk=Hk)
    cq_d, ck_d = torch.tensor(cq, dtype=torch.int32).cuda(), torch.tensor(ck, dtype=torch.int32).cuda()
    out = torch.full((sum(lens_q), H, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    from liteattention_amd.flash_attn_interface import mha_fwd
    o, lse, *_ = mha_fwd(q.cuda(), k.cuda(), v.cuda(), out=out, cu_seqlens_q=cq_d, cu_seqlens_k=ck_d,
                         max_seqlen_q=max(lens_q), max_seqlen_k=max(lens_k))
    assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
    for b in range(len(lens_q)):
        sl_q, sl_k = slice(cq[b], cq[b + 1]), slice(ck[b], ck[b + 1])
        o_r, lse_r, _ = orc.qkskip_fwd(q[sl_q][None], k[sl_k][None], v[sl_k][None], block_m=bm, block_n=bn,
                                       softmax_scale=D ** -0.5)
        # 0.75 ulp of the largest output (tests/test_gpu_head_dims.py::test_dense_matches_oracle has the accounting: half an ulp for
        # the bf16 store + P rounded relative to the lazily updated reference max, not averaged away over a 65-key sequence)
        assert (o[sl_q].float().cpu() - o_r[0]).abs().max().item() <= 1.5 * 2.0 ** -8 * o_r.abs().max().item() + 1e-3, (D, b)
        assert (lse[:, sl_q].cpu() - lse_r[0]).abs().max().item() <= 1e-3, (D, b)
    # the L.flash_attn_varlen_func wrapper: same launch, host lists accepted, max_seqlen derived from them
    o2 = L.flash_attn_varlen_func(q.cuda(), k.cuda(), v.cuda(), cq, ck)
    assert torch.equal(o2, o)
    with pytest.raises(NotImplementedError):
        L.flash_attn_varlen_func(q.cuda(), k.cuda(), v.cuda(), cq, ck, seqused_k=torch.ones(6, device="cuda"))
    with pytest.raises(RuntimeError):
        L.flash_attn_varlen_func(q.cuda(), k.cuda(), v.cuda(), cq_d, ck_d)          # device cu_seqlens need max_seqlen_*


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
@pytest.mark.parametrize("Sk,H,Hk", [(1, 1, 1), (3, 2, 1), (7, 3, 3), (12, 1, 1), (13, 4, 2), (63, 2, 2)])
def test_key_sequences_shorter_than_a_dma_piece(dtype, Sk, H, Hk):
    """Regression (found by the varlen tests): the hand-scheduled loop issues its LDS-DMA pieces with the piece index on the
    instruction offset and compensates it in the per-lane offset; with rows clamped to a very short sequence that offset went
    negative = +4 GiB in the unsigned SADDR form (memory fault, or a silent read of masked garbage). MHA / GQA / MQA row strides."""
    import liteattention_amd as L
    from oracle import oracle as orc
    es = 1 if dtype == "fp8" else 2
    bm, bn = _tiles(128, es)
    g = torch.Generator().manual_seed(Sk * 31 + H)
    cast = (lambda x: x.to(torch.float8_e4m3fn)) if dtype == "fp8" else (lambda x: x.bfloat16())
    q = cast(torch.randn(2, 300, H, 128, generator=g))
    k = cast(torch.randn(2, Sk, Hk, 128, generator=g))
    v = cast(torch.randn(2, Sk, Hk, 128, generator=g))
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=fp8_p_round() if dtype == "fp8" else True)
    tol = (0.05 * o_ref.abs().max().item() + 2e-2) if dtype == "fp8" else _tol(o_ref)
    assert (out.float().cpu() - o_ref).abs().max().item() <= tol
    assert (lse.cpu() - lse_ref).abs().max().item() <= (fp8_lse_tol() if dtype == "fp8" else 1e-3)


@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("fp8", 128), ("bf16", 192), ("bf16", 64)])
@pytest.mark.parametrize("Sk", [1, 13, 64])
def test_a_single_key_tile_through_the_class(dtype, D, Sk):
    """Key sequences of at most one k-tile through ``LiteAttention.__call__`` (lists walked, dynamic work distribution), three steps.
    The list rows are 2 ints wide there - too short for the one range [0, 0] they describe ([len, start, end]; the reference's format
    has no room for it and its writer stores past the row): the reader takes the missing end as 0, the writer counts but does not
    store the entry behind the row (kernels and oracle alike). Regression of round 3 (tools/fuzz_parity.py): the host raised
    'must_do_list has more entries than k tiles' for the default must-do row, so ``LiteAttention`` could not be called at all."""
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    es = 1 if dtype == "fp8" else 2
    bm, bn = _tiles(D, es)
    B, Sq, H, thr = 2, 300, 2, -1.0
    Qt, Kt = -(-Sq // bm), 1
    cast = (lambda x: x.to(torch.float8_e4m3fn)) if dtype == "fp8" else (lambda x: x.bfloat16())
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    md_row = orc.expand_must_do_ref([0, 0], bn, 3)
    margins = torch.empty(B, H, Qt, Kt)
    for step in range(3):
        g = torch.Generator().manual_seed(Sk * 7 + step)
        q, k, v = cast(torch.randn(B, Sq, H, D, generator=g)), cast(torch.randn(B, Sk, H, D, generator=g)), cast(torch.randn(B, Sk, H, D, generator=g))
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        assert att._skip_list.shape[-1] == 2
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, must_do_list=md_row, thr=thr,
                                                 margins=margins, p_round=fp8_p_round() if dtype == "fp8" else True)
        assert n_tiles == B * H * Qt
        tol = (0.05 * o_ref.abs().max().item() + 2e-2) if dtype == "fp8" else 2.0 ** -7 * o_ref.abs().max().item() + 1e-3
        assert (out.float().cpu() - o_ref).abs().max().item() <= tol
        assert (lse.cpu() - lse_ref).abs().max().item() <= (fp8_lse_tol() if dtype == "fp8" else 1e-3)
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0 and torch.equal(wr[..., 1], torch.zeros_like(wr[..., 1]))


# ------------------------------------------------------------------------------------------ f3: must_skip_list
@pytest.mark.parametrize("with_must_do", [False, True])
def test_must_skip_list_through_the_kernel(with_must_do):
    """README.md:193-197 / lite_attention.py:126-145: lists INITIALISED from ``must_skip_list`` token ranges, then three
    denoising-like steps. Outputs and write lists against the oracle run from the same initial lists; the dropped tiles are
    never walked again (a tile absent from the read list cannot reappear in the write list)."""
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    B, S, H, thr = 2, 2048, 2, -4.0
    Qt, Kt = S // BM, S // BN
    must_skip = [1500, 1100, 600, 250]                       # two token ranges [start, end), descending
    must_do = [1200, 1000] if with_must_do else None         # overlaps the first skipped range: has nothing left to protect there
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    md_row = orc.expand_must_do_ref(must_do if with_must_do else [0, 0], BN, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    dropped = None
    for step in range(3):
        q, k, v = structured_qkv(B, S, H, 128, seed=400)
        g = torch.Generator().manual_seed(4000 + step)
        q = (q.float() + 0.05 * torch.randn(q.shape, generator=g)).bfloat16()
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, must_do_list=must_do, must_skip_list=must_skip)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        if step == 0:
            row = rd[0, 0, 0].tolist()
            walked = orc.walk_tiles(row)
            dropped = sorted(set(range(Kt)) - set(walked))
            # the reference's conversion (lite_attention.py:129-138): a skipped range's start rounds UP, its end DOWN, to the
            # tiles that stay listed: listed 31..24, 17..10, 3..0 -> dropped 18..23 and 4..9
            assert dropped == list(range(4, 10)) + list(range(18, 24)), dropped
            assert (rd[:, :, :, : row[0] + 1] == torch.tensor(row[: row[0] + 1], dtype=torch.int32)).all()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc,
                                                 must_do_list=md_row, thr=thr, margins=margins)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0
        for b in range(B):
            for h in range(H):
                for m in range(Qt):
                    assert not (set(orc.walk_tiles(wr[b, h, m].tolist())) & set(dropped))
    assert orc.listed_tiles(att.current_read_list()[:B].cpu()) < B * H * Qt * (Kt - len(dropped))      # and QK-Skip dropped more


def test_must_do_list_that_fills_the_whole_row():
    """a multi-range must-do list with len >= k_tiles - 1 made the serial writer read 1-2 ints past its row (the
    reference has the same flaw, mainloop...:156-159). The reader now treats a pair outside the row as (0, 0), in the kernel and
    in the oracle alike. Detector: in the 4-D tensor the ints behind q-tile 0's row are q-tile 1's row, crafted so that an
    over-read pair (8, -1) would make tile 0 must-do; at thr = +inf every tile that is not must-do is dropped."""
    import liteattention_amd as L
    from oracle import oracle as orc
    B, S, H = 1, 512, 1
    Qt, Kt = math.ceil(S / BM), S // BN                       # Kt = 8: four ranges = 8 entries fill the row
    must_do = [512, 448, 384, 320, 256, 192, 128, 64]         # tile ranges (8,7] (6,5] (4,3] (2,1]: must-do tiles 6, 4, 2
    q, k, v = structured_qkv(B, S, H, 128, seed=77, alpha=9.0)
    lists = L.LiteAttention.init_skip_list(B, S, H, 128, False, torch.bfloat16, "cuda")
    md4 = L.LiteAttention._expand_must_do_list(must_do, (B, H, Qt, Kt + 1), q.cuda(), v.cuda())
    assert int(md4[0, 0, 0, 0]) == Kt                         # the list fills the row completely
    md4[0, 0, 1:] = 0
    md4[0, 0, 1:, 0] = 8                                      # rows of the other q-tiles: [8, -1, 0, ...] = ranges that match nothing,
    md4[0, 0, 1:, 1] = -1                                     # but (start 8, end -1) when read as a PAIR from the row before
    L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), attn_read_list=lists[0], attn_must_do_list=md4, attn_write_list=lists[1],
                      thr=float("inf"))
    wr_orc = torch.zeros_like(lists[1].cpu())
    orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=lists[0].cpu(), write_list=wr_orc, must_do_list=md4.cpu(),
                   thr=float("inf"))
    wr = lists[1].cpu()
    # first tile, the must-do tiles 6 / 4 / 2, and after each of them the first flagged tile of the run (the writer lists the tile
    # at which a skipped run starts, mainloop...:163-168); tile 0 is NOT listed: it would be if the pair behind the row were read
    assert orc.walk_tiles(wr[0, 0, 0].tolist()) == [7, 6, 5, 4, 3, 2, 1]
    for m in range(Qt):
        n = int(wr_orc[0, 0, m, 0])
        assert wr[0, 0, m, : n + 1].tolist() == wr_orc[0, 0, m, : n + 1].tolist()


# ------------------------------------------------------------------------------------------ f4: checkpoint / resume
@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_checkpoint_resume_is_bit_identical_on_the_device(dtype):
    """SURVEY §5 checkpoint/resume: 6 steps straight == 3 steps -> state_dict() -> NEW object -> load_state_dict() -> 3 steps,
    outputs and both list buffers bit-identical. load_state_dict() without device= restores to the device the state came from
    (round 1 left the lists on the CPU and the next call silently re-initialised them)."""
    import liteattention_amd as L
    B, S, H, thr = 1, 3072, 3, -3.0

    def inputs(step):
        q, k, v = structured_qkv(B, S, H, 128, seed=500, alpha=9.0 if dtype == "fp8" else 10.0,
                                 dtype=torch.float32 if dtype == "fp8" else torch.bfloat16)
        g = torch.Generator().manual_seed(5000 + step)
        q = q.float() + 0.05 * torch.randn(q.shape, generator=g)
        cast = (lambda x: x.to(torch.float8_e4m3fn)) if dtype == "fp8" else (lambda x: x.bfloat16())
        return cast(q).cuda(), cast(k.float()).cuda(), cast(v.float()).cuda()

    straight = L.LiteAttention(threshold=thr, max_batch_size=B)
    outs = [straight(*inputs(s)) for s in range(6)]
    first = L.LiteAttention(threshold=thr, max_batch_size=B)
    for s in range(3):
        assert torch.equal(first(*inputs(s)), outs[s])
    state = first.state_dict()
    assert state["skip_list"].device.type == "cpu" and state["device"].startswith("cuda")
    state = {k_: (v_.clone() if isinstance(v_, torch.Tensor) else v_) for k_, v_ in state.items()}      # as if read back from disk
    del first
    resumed = L.LiteAttention()                                # default threshold / batch size: everything comes from the state
    resumed.load_state_dict(state)
    assert resumed._skip_list.is_cuda and resumed.threshold == thr and resumed._phase == 1
    for s in range(3, 6):
        out = resumed(*inputs(s))
        assert torch.equal(out, outs[s]), s
    assert torch.equal(resumed._skip_list, straight._skip_list) and resumed._phase == straight._phase
    assert 0.02 < resumed.get_skip_fraction(batch=B) < 0.98    # the run really skipped (and the state carried it)
    # a state without a recorded device (a round-1 checkpoint) must name one
    legacy = dict(state)
    legacy.pop("device")
    with pytest.raises(ValueError, match="device"):
        L.LiteAttention().load_state_dict(legacy)
    again = L.LiteAttention()
    again.load_state_dict(legacy, device="cuda")
    assert again._shape_key[5] == inputs(0)[0].device


# ------------------------------------------------------------------------------------------ boundary (ABI 4)
def _raw_args(q, k, v, out, lse):
    from liteattention_amd import _cabi
    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.dtype = _cabi.LA_DTYPE_BF16
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr()
    for n, t in (("q", q), ("k", k), ("v", v), ("o", out)):
        setattr(a, f"{n}_batch_stride", t.stride(0)); setattr(a, f"{n}_row_stride", t.stride(1)); setattr(a, f"{n}_head_stride", t.stride(2))
    a.batch, a.seqlen_q, a.seqlen_k = q.shape[0], q.shape[1], k.shape[1]
    a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = q.shape[2], k.shape[2], q.shape[3], q.shape[3]
    a.softmax_scale = q.shape[3] ** -0.5
    return a


def test_empty_key_sequence_is_handled_inside_the_library():
    """flash_api.cpp:1241-1245: seqlen_k == 0 -> o = 0, lse = +inf. A non-Python host calling la_fwd gets the reference's RESULT,
    not an error code; strided `o` (a head slice of a wider tensor) is respected, neighbours untouched."""
    from liteattention_amd import _cabi
    lib = _cabi.load()
    q = torch.randn(2, 37, 3, 128, device="cuda").bfloat16()
    k = torch.zeros(2, 0, 3, 128, dtype=torch.bfloat16, device="cuda")
    wide = torch.full((2, 37, 5, 128), 7.0, dtype=torch.bfloat16, device="cuda")
    out = wide[:, :, 1:4]
    lse = torch.zeros(2, 3, 37, device="cuda")
    a = _raw_args(q, k, k, out, lse)
    a.k = a.v = q.data_ptr()                                  # any non-NULL aligned pointer: never dereferenced
    a.block_m, a.block_n = BM, BN
    a.flags = _cabi.default_flags() & _cabi.GEOMETRY_FLAGS     # (BM, BN) is the tile of the kernel the environment selects
    rc = lib.la_fwd(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == _cabi.LA_OK, _cabi.status_string(rc)
    torch.cuda.synchronize()
    assert (out == 0).all() and torch.isinf(lse).all() and (lse > 0).all()
    assert (wide[:, :, 0] == 7).all() and (wide[:, :, 4] == 7).all()
    # and through the Python surface (fp8 too: the library answers before it asks for a workspace)
    import liteattention_amd as L
    o2, l2 = L.flash_attn_func(q, k, k, return_softmax_lse=True)
    assert (o2 == 0).all() and torch.isinf(l2).all() and (l2 > 0).all()
    o8 = L.flash_attn_func(q.to(torch.float8_e4m3fn), k.to(torch.float8_e4m3fn), k.to(torch.float8_e4m3fn))
    assert o8.dtype == torch.bfloat16 and (o8 == 0).all()


def test_kernel_selection_flags_through_the_raw_cabi():
    """LA_FLAG_KERNEL_128ROW runs the 128-row kernel on lists of 128-row q-tiles; LA_FLAG_EXACT_RESCALE (tau = 0) changes the
    rescale schedule, not the lists. Both against the oracle at the tile la_get_tile_sizes_ex reports."""
    from liteattention_amd import _cabi
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    lib = _cabi.load()
    B, S, H, thr = 1, 1536, 2, -3.0
    q, k, v = structured_qkv(B, S, H, 128, seed=600)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    results = {}
    for flags in (0, _cabi.LA_FLAG_KERNEL_128ROW, _cabi.LA_FLAG_EXACT_RESCALE, _cabi.LA_FLAG_STATIC_SCHED):
        m, n = ctypes.c_int(), ctypes.c_int()
        assert lib.la_get_tile_sizes_ex(128, 2, flags, ctypes.byref(m), ctypes.byref(n)) == 0
        bm, bn = m.value, n.value
        assert bm == (128 if flags == _cabi.LA_FLAG_KERNEL_128ROW else 256)
        Qt, Kt = S // bm, S // bn
        from liteattention_amd import skip_lists as sl
        lists = sl.new_skip_lists(B, H, Qt, Kt, "cuda")
        out = torch.empty_like(qd)
        lse = torch.empty(B, H, S, device="cuda")
        ws = torch.zeros(1024, dtype=torch.uint8, device="cuda")
        margins = torch.empty(B, H, Qt, Kt)
        for step in range(2):
            a = _raw_args(qd, kd, vd, out, lse)
            a.block_m, a.block_n, a.flags, a.thr = bm, bn, flags, thr
            a.read_list, a.write_list = lists[step % 2].data_ptr(), lists[1 - step % 2].data_ptr()
            a.workspace, a.workspace_bytes = ws.data_ptr(), 1024
            rc = lib.la_fwd(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == _cabi.LA_OK, _cabi.status_string(rc)
            torch.cuda.synchronize()
            rd, wr = lists[step % 2].cpu(), lists[1 - step % 2].cpu()
            wr_orc = torch.zeros_like(wr)
            o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, thr=thr,
                                               margins=margins)
            assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
            assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
            bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
            assert bad == 0
        results[flags] = (out.clone(), lists.clone())
    assert torch.equal(results[0][1], results[_cabi.LA_FLAG_EXACT_RESCALE][1])            # same lists, lazily or exactly rescaled
    assert torch.equal(results[0][0], results[_cabi.LA_FLAG_STATIC_SCHED][0])             # dynamic == static bit-exactly
    assert torch.equal(results[0][1], results[_cabi.LA_FLAG_STATIC_SCHED][1])


# ------------------------------------------------------------------------------------------ race screen
@pytest.mark.parametrize("path", ["dynamic", "windows_after_first_static"])
def test_race_screen_1000_iterations(path):
    """The reference's race screen (hopper/tests/test_flash_attn.py:1144-1175: 1000 launches under memory pressure, all results
    identical to the first) on the two paths collectives interleave with: the dynamic ticket queues (persistent workgroups,
    stealing) and the windowed form with per-item workgroups after the first window. 640 (head, q-tile) items > 256 CUs, real
    (fragmented, unequal) lists, a co-running memory hog on a second stream."""
    import liteattention_amd as L
    B, S, H, thr = 1, 10240, 16, -3.0
    q, k, v = [x.cuda() for x in structured_qkv(B, S, H, 128, seed=700, frames=16)]
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    for _ in range(3):                                          # get to a real, fragmented list
        att(q, k, v)
    base = att._skip_list.clone()
    phase = att._phase
    frac = att.get_skip_fraction(batch=B)
    assert 0.05 < frac < 0.95, frac
    qt = -(-S // BM)
    windows = [(0, qt // 3), (qt // 3, qt // 3), (2 * (qt // 3), qt - 2 * (qt // 3))]

    def run():
        att._skip_list.copy_(base)
        att._phase = phase
        if path == "dynamic":
            return att(q, k, v, return_softmax_lse=True)
        return att.call_windowed(q, k, v, windows, return_softmax_lse=True, static_sched="after_first")

    hog_stream = torch.cuda.Stream()
    hog_a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")      # 256 MiB copies beside the kernel: memory pressure
    hog_b = torch.empty_like(hog_a)
    out0, lse0 = run()
    out0, lse0, lists0 = out0.clone(), lse0.clone(), att._skip_list.clone()
    for it in range(1000):
        if it % 4 == 0:
            with torch.cuda.stream(hog_stream):
                hog_b.copy_(hog_a)
        out, lse = run()
        assert torch.equal(out, out0) and torch.equal(lse, lse0) and torch.equal(att._skip_list, lists0), it
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------ malformed lists
@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("fp8", 128), ("bf16", 64), ("bf16", 96), ("bf16", 192), ("bf16", 256)])
def test_garbage_read_lists_are_memory_safe(dtype, D):
    """The read list is caller-owned memory: whatever it holds (negative or huge tile numbers, ascending or overlapping ranges,
    a length word beyond the row, zeros), the kernel must stay inside q / k / v / the write list - tile numbers are clamped to
    [0, Kt), the walk to Kt positions (la_fwd_common.h expand_read_list) - and produce finite numbers. Ragged Sq and Sk, so the
    clamped last tile turns up at positions other than the first. The write list sits between guard words."""
    import liteattention_amd as L
    es = 1 if dtype == "fp8" else 2
    bm, bn = _tiles(D, es)
    B, Sq, Sk, H = 2, 1000, 1250, 3
    qt, kt = math.ceil(Sq / bm), math.ceil(Sk / bn)
    g = torch.Generator().manual_seed(D + es)
    cast = (lambda x: x.to(torch.float8_e4m3fn)) if dtype == "fp8" else (lambda x: x.bfloat16())
    q, k, v = [cast(torch.randn(B, s_, H, D, generator=g)).cuda() for s_ in (Sq, Sk, Sk)]
    rows = B * H * qt
    kinds = [torch.randint(-2 ** 31, 2 ** 31 - 1, (rows, kt + 1), generator=g, dtype=torch.int64),       # anything at all
             torch.randint(-5, kt + 5, (rows, kt + 1), generator=g, dtype=torch.int64),                  # nearly valid numbers, no order
             torch.zeros(rows, kt + 1, dtype=torch.int64)]                                               # len 0, all tile 0
    asc = torch.arange(kt + 1).repeat(rows, 1)                                                           # ascending "ranges", len = Kt + 7
    asc[:, 0] = kt + 7
    kinds.append(asc)
    for read in kinds:
        read = read.to(torch.int32).view(B, H, qt, kt + 1).cuda()
        guard = torch.full((rows * (kt + 1) + 64,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        write = guard[32: 32 + rows * (kt + 1)].view(B, H, qt, kt + 1)
        for thr in (-2.0, float("inf")):
            out, lse = L.flash_attn_func(q, k, v, attn_read_list=read, attn_write_list=write, thr=thr, return_softmax_lse=True)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(out.float()).all()) and not bool(torch.isnan(lse).any())
            assert bool((guard[:32] == 0x5A5A5A5A).all()) and bool((guard[-32:] == 0x5A5A5A5A).all())
            assert int(write[..., 0].min()) >= 0 and int(write[..., 0].max()) <= kt
