"""GPU: head dims 64, 96, 192 and 256 on the hand-scheduled kernel. 64 (round 3): the head_dim-128 form (q-tile 256) with 8 of the 16 K /
V^T fragments per tile on an LDS image of 128-byte rows (generator run with LA_X64_D=64). 256: 32 query rows per wave, q-tile 128 x k-tile 64 (generator run with
LA_X64_D=256); 192: the same with 24 of the 32 K / V^T fragments per tile; 96: the head_dim-128 form (q-tile 256) with 12 of 16.
The reference builds these head sizes by default (hopper/setup.py:57-61, instantiations/flash_fwd_hdim{96,192,256}_bf16_sm90.cu). Cases:
dense ragged shapes and skip lists over several steps against the oracle, bf16 and fp16; the persistent multi-item loop with ticket
stealing (more items than CUs) dynamic == static; LA_FLAG_KERNEL_128ROW (the hipcc-scheduled instantiation) as an independent
second implementation with the same tiles; the longest supported key sequence."""
import os

import pytest
import torch

from helpers import structured_qkv
from test_gpu_parity import _compare_lists

pytestmark = pytest.mark.gpu
DIMS = [64, 96, 192, 256]


def _L():
    import liteattention_amd as L
    small = (128, 64) if os.environ.get("LA_VOTE", "").startswith("half") else (256, 64)      # LA_FLAG_HALF_VOTE: lists per 128-row half at head dims <= 128
    assert L.get_tile_sizes(256, 2) == (128, 64) and L.get_tile_sizes(192, 2) == (128, 64) and L.get_tile_sizes(96, 2) == L.get_tile_sizes(64, 2) == small
    from liteattention_amd.flash_attn_interface import kernel_head_dim
    assert [kernel_head_dim(d, 2) for d in (40, 64, 72, 96, 160, 192, 256)] == [64, 64, 96, 96, 192, 192, 256]      # native, not zero-padded
    return L


def _orc():
    from oracle import oracle as orc
    return orc


def _randn(B, Sq, Sk, H, seed, dtype=torch.bfloat16, Hk=None, D=256):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, Sq, H, D, generator=g).to(dtype), torch.randn(B, Sk, Hk or H, D, generator=g).to(dtype),
            torch.randn(B, Sk, Hk or H, D, generator=g).to(dtype))


def _tol(o_ref, dtype=torch.bfloat16, ulps=0.5):
    """`ulps` units in the last place of the largest output (bf16 8 bits, fp16 11) + 1e-3."""
    return ulps * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10) * o_ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (1, 17, 17, 1), (2, 129, 65, 3), (1, 1000, 1000, 2), (1, 300, 2100, 2), (1, 128, 64, 1),
                                   (2, 257, 640, 4)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", DIMS)
def test_dense_matches_oracle(shape, dtype, D):
    L, orc = _L(), _orc()
    B, Sq, Sk, H = shape
    q, k, v = _randn(B, Sq, Sk, H, seed=Sq * 7 + Sk, dtype=dtype, D=D)
    out = torch.full((B, Sq, H, D), float("nan"), dtype=dtype, device="cuda")
    from liteattention_amd.flash_attn_interface import mha_fwd
    o2, lse, *_ = mha_fwd(q.cuda(), k.cuda(), v.cuda(), out=out)
    assert o2 is out and torch.isfinite(out.float()).all()
    bm, bn = L.get_tile_sizes(D, 2)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round="f16" if dtype == torch.float16 else True)
    # half an ulp for the 16-bit store of O + a quarter for P rounded relative to the lazy reference max instead of the oracle's running
    # max (not averaged away over the few keys of the short shapes; measured worst case over this grid: 0.62 ulp at Sk = 65)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref, dtype, ulps=0.75)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("D", DIMS)
def test_gqa_and_strided_inputs(D):
    """K embedded in a wider tensor (rows of 2 D elements, of which the kernel may read only the first D: the DMA lanes that have no
    column of their own at head dims 96 / 192 must stay inside the row - the other half is NaN here)."""
    L, orc = _L(), _orc()
    q, k, v = _randn(2, 200, 333, 6, seed=3, Hk=2, D=D)
    big = torch.full((2, 333, 2, 2 * D), float("nan"), dtype=torch.bfloat16)
    big[..., :D] = k
    ks = big.cuda()[..., :D]
    out, lse = L.flash_attn_func(q.cuda(), ks, v.cuda(), softmax_scale=0.05, return_softmax_lse=True)
    bm, bn = L.get_tile_sizes(D, 2)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, softmax_scale=0.05)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("use_must_do", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", DIMS)
def test_multi_step_lists_match_oracle(use_must_do, dtype, D):
    L, orc = _L(), _orc()
    thr = -2.0
    B, S, H = 2, 1536, 2
    BM, BN = L.get_tile_sizes(D, 2)
    Qt, Kt = S // BM, S // BN
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    must_do = [700, 400] if use_must_do else None
    md_row = orc.expand_must_do_ref(must_do if use_must_do else [0, 0], BN, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    total_border, listed = 0, []
    for step in range(5):
        q, k, v = structured_qkv(B, S, H, D, seed=100, alpha=7.0, dtype=torch.float32)
        g = torch.Generator().manual_seed(1000 + step)
        q = (q + 0.05 * torch.randn(q.shape, generator=g)).to(dtype)
        k = (k + 0.05 * torch.randn(k.shape, generator=g)).to(dtype)
        v = v.to(dtype)
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, must_do_list=must_do)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc, must_do_list=md_row,
                                                 thr=thr, margins=margins, p_round="f16" if dtype == torch.float16 else True)
        # structured inputs at head_dim 256 give rows with one or two dominant keys: P of such a key is rounded to 16 bits relative to
        # the lazily updated reference max here and to the true running max in the oracle - different roundings of the same weight,
        # 2^-9 relative each, not averaged away (measured: <= 0.9 ulp of the largest output; the hipcc-scheduled kernel, which
        # rescales every step like the oracle, <= 0.5). One ulp is the bound; LSE and the lists do not see P's rounding.
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref, dtype, ulps=1.0)
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
        bad, border = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0, f"step {step}: {bad} rows differ from the oracle with no borderline tile"
        total_border += border
        listed.append(orc.listed_tiles(wr[:B]))
        assert n_tiles == orc.listed_tiles(rd[:B])
    assert listed == sorted(listed, reverse=True) and listed[-1] < 0.97 * B * H * Qt * Kt     # tiles really were dropped
    assert total_border <= 2


@pytest.mark.parametrize("D", DIMS)
def test_persistent_loop_with_stealing_dynamic_equals_static(D):
    """B * H * q_tiles = 1024 (head_dim 96: 512) items on 256 CUs, lists of different lengths per q-tile: the ticket path (re-iteration, stealing between
    the per-XCD queues) must give bit-identical outputs and lists to the static one-workgroup-per-item map."""
    L = _L()
    B, S, H = 1, 8192, 16
    q, k, v = [x.cuda() for x in structured_qkv(B, S, H, D, seed=7, alpha=7.0)]
    res = []
    for static in (False, True):
        att = L.LiteAttention(threshold=-2.0, max_batch_size=B)
        outs = []
        for step in range(3):
            from liteattention_amd.flash_attn_interface import mha_fwd
            rd, wr = att._get_read_write_lists(q, v)
            o, lse, *_ = mha_fwd(q, k, v, attn_read_list=rd, attn_write_list=wr, thr=att.threshold, _static_sched=static)
            outs.append((o.clone(), lse.clone(), wr.clone()))
        res.append(outs)
    for (o1, l1, w1), (o2, l2, w2) in zip(*res):
        assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(w1, w2)
    bm, bn = L.get_tile_sizes(D, 2)
    assert L.skip_list_stats(res[0][-1][2], 1)[0].item() < 0.95 * H * (S // bm) * (S // bn)


def test_hand_scheduled_and_hipcc_scheduled_d256_kernels_agree(monkeypatch):
    D = 256
    """LA_FLAG_KERNEL_128ROW selects the hipcc-scheduled 128-row template at head_dim 256 (same tiles): a second implementation of
    the same walk. Write lists must be identical; outputs differ only by the lazy rescale (tau = 8 vs every step)."""
    L = _L()
    B, S, H = 1, 2048, 3
    q, k, v = [x.cuda() for x in structured_qkv(B, S, H, D, seed=11, alpha=7.0)]
    runs = {}
    for name in ("x64", "v2"):
        if name == "v2":
            monkeypatch.setenv("LA_FWD_KERNEL", "v2")
        att = L.LiteAttention(threshold=-2.0, max_batch_size=B)
        for _ in range(3):
            o, lse = att(q, k, v, return_softmax_lse=True)
        runs[name] = (o, lse, att._skip_list.clone())
    assert torch.equal(runs["x64"][2], runs["v2"][2])
    assert (runs["x64"][0].float() - runs["v2"][0].float()).abs().max().item() <= 2.0 ** -7 * runs["v2"][0].float().abs().max().item()
    assert (runs["x64"][1] - runs["v2"][1]).abs().max().item() <= 1e-4


@pytest.mark.parametrize("D", [192, 256])
def test_longest_key_sequence(D):
    """128 KiB of the 160 KiB LDS are K/V rings at head dims 192 / 256: 20.25 bytes per key tile for the walk leave ~1 570 tiles (100 k keys)."""
    L = _L()
    from liteattention_amd import _cabi
    g = torch.Generator(device="cuda").manual_seed(0)
    Sk = 1500 * 64
    q = torch.randn(1, 128, 1, D, device="cuda", generator=g).bfloat16()
    k = torch.randn(1, Sk, 1, D, device="cuda", generator=g).bfloat16()
    v = torch.randn(1, Sk, 1, D, device="cuda", generator=g).bfloat16()
    out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    ref = torch.nn.functional.scaled_dot_product_attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2))
    assert (out.float() - ref.transpose(1, 2)).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-3
    # beyond one launch's walk (round 6): a DENSE call is cut into runs of tiles inside la_fwd and merged by LSE; a call with skip lists
    # keeps the bound and its typed error
    Sk2 = 2048 * 64 - 5                                     # S = 131 072: two runs
    q = torch.randn(1, 131, 1, D, device="cuda", generator=g).bfloat16()       # an odd row count: the partial LSE buffers of the runs are not 16-byte multiples
    k2 = torch.randn(1, Sk2, 1, D, device="cuda", generator=g).bfloat16()
    v2 = torch.randn(1, Sk2, 1, D, device="cuda", generator=g).bfloat16()
    out2, lse2 = L.flash_attn_func(q, k2, v2, return_softmax_lse=True)
    sc = q.float()[0, :, 0] @ k2.float()[0, :, 0].T / D ** 0.5
    ref2 = torch.softmax(sc, -1) @ v2.float()[0, :, 0]
    assert (out2.float()[0, :, 0] - ref2).abs().max().item() <= 2.0 ** -7 * ref2.abs().max().item() + 1e-3      # two bf16 roundings: partials, merge
    assert (lse2[0, 0] - torch.logsumexp(sc, -1)).abs().max().item() <= 1e-3
    bm, bn = L.get_tile_sizes(D, 2)
    lists = torch.zeros(2, 1, 1, -(-131 // bm), -(-Sk2 // bn) + 1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match=_cabi.status_string(_cabi.LA_ERR_SEQLEN)[:20]):
        L.flash_attn_func(q, k2, v2, attn_read_list=lists[0], attn_write_list=lists[1])


@pytest.mark.parametrize("D", DIMS)
def test_race_screen_200_iterations(D):
    """The reference's race screen (hopper/tests/test_flash_attn.py:1144-1175; the 1000-iteration form runs at head_dim 128 in
    test_gpu_round2.py) on these bodies: real, fragmented, unequal lists, more items than CUs (persistent loop, ticket stealing), a
    co-running memory hog on a second stream; every launch identical to the first in O, LSE and the write list."""
    L = _L()
    B, S, H, thr = 1, 8192, 8, -2.0
    q, k, v = [x.cuda() for x in structured_qkv(B, S, H, D, seed=700, alpha=7.0, frames=16)]
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    for _ in range(3):
        att(q, k, v)
    base, phase = att._skip_list.clone(), att._phase
    assert 0.02 < att.get_skip_fraction(batch=B) < 0.95

    def run():
        att._skip_list.copy_(base)
        att._phase = phase
        return att(q, k, v, return_softmax_lse=True)

    hog_stream = torch.cuda.Stream()
    hog_a = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    hog_b = torch.empty_like(hog_a)
    out0, lse0 = run()
    out0, lse0, lists0 = out0.clone(), lse0.clone(), att._skip_list.clone()
    for it in range(200):
        if it % 4 == 0:
            with torch.cuda.stream(hog_stream):
                hog_b.copy_(hog_a)
        out, lse = run()
        assert torch.equal(out, out0) and torch.equal(lse, lse0) and torch.equal(att._skip_list, lists0), it
    torch.cuda.synchronize()
