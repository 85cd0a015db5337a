"""GPU: the adapter surfaces against the CPU oracle."""
import math

import pytest
import torch

from helpers import structured_qkv

pytestmark = pytest.mark.gpu


def _tiles():
    import liteattention_amd as L
    return L.get_tile_sizes(128, 2)


BM, BN = _tiles()       # the lists follow the selected kernel's tile


def _tol(o):
    return 2.0 ** -8 * o.abs().max().item() + 1e-3


def test_fa2_surface_and_varlen_match_oracle():
    import liteattention_amd as L
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(3)
    lens_q, lens_k = [300, 77, 512], [512, 200, 64]        # cross-attention-like: Sq != Sk per sequence
    H, D = 4, 128
    q = torch.randn(sum(lens_q), H, D, generator=g).bfloat16()
    k = torch.randn(sum(lens_k), H, D, generator=g).bfloat16()
    v = torch.randn(sum(lens_k), H, D, generator=g).bfloat16()
    cq = [0] + torch.tensor(lens_q).cumsum(0).tolist()
    ck = [0] + torch.tensor(lens_k).cumsum(0).tolist()
    out, lse, _ = L.flash_attn_varlen_func(q.cuda(), k.cuda(), v.cuda(), torch.tensor(cq, dtype=torch.int32).cuda(),
                                           torch.tensor(ck, dtype=torch.int32).cuda(), max(lens_q), max(lens_k),
                                           return_attn_probs=True)
    for b in range(3):
        o_ref, lse_ref, _ = orc.qkskip_fwd(q[cq[b]:cq[b + 1]][None], k[ck[b]:ck[b + 1]][None], v[ck[b]:ck[b + 1]][None],
                                           block_m=BM, block_n=BN)
        assert (out[cq[b]:cq[b + 1]].float().cpu() - o_ref[0]).abs().max().item() <= _tol(o_ref)
        assert (lse[:, cq[b]:cq[b + 1]].cpu() - lse_ref[0]).abs().max().item() <= 1e-3
    o2 = L.fa2_flash_attn_func(q[:300][None].cuda(), k[:512][None].cuda(), v[:512][None].cuda(), softmax_scale=0.07)
    o_ref, _, _ = orc.qkskip_fwd(q[:300][None], k[:512][None], v[:512][None], block_m=BM, block_n=BN, softmax_scale=0.07)
    assert (o2.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)


def test_static_block_mask_matches_masked_dense_attention():
    import liteattention_amd as L
    from oracle import oracle as orc
    B, S, H = 2, 1100, 2                                       # ragged: 9 q-tiles, 18 k-tiles, last ones partial
    Qt, Kt = math.ceil(S / BM), math.ceil(S / BN)
    g = torch.Generator().manual_seed(5)
    q, k, v = [torch.randn(B, S, H, 128, generator=g).bfloat16() for _ in range(3)]
    mask = torch.rand(Qt, Kt, generator=g) < 0.4
    mask[:, Kt - 1] = True                                     # keep the ragged tile for every row (first walked)
    mask[3] = False
    mask[3, 5] = True                                          # a row with a single interior tile
    lists = L.blockmask_to_skip_lists(mask, B, H, "cuda")
    out, lse = L.flash_blocksparse_attn_func(q.cuda(), k.cuda(), v.cuda(), mask, return_softmax_lse=True, skip_lists=lists)
    # oracle 1: tiled oracle with the same lists
    wr = torch.zeros_like(lists[1].cpu())
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=lists[0].cpu(), write_list=wr,
                                       thr=float("-inf"))
    assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    # oracle 2: eager attention with -inf on dropped blocks
    full = mask.repeat_interleave(BM, 0)[:S].repeat_interleave(BN, 1)[:, :S]
    s = torch.einsum("bthd,bshd->bhts", q.float() * 128 ** -0.5, k.float()).masked_fill(~full, float("-inf"))
    ref = torch.einsum("bhts,bshd->bthd", torch.softmax(s, -1), v.float())
    assert (out.float().cpu() - ref).abs().max().item() <= 3e-2
    assert torch.equal(lists[0], lists[1])                     # thr=-inf: the static pattern is a fixed point


def test_threshold_calibration_hits_target_sparsity():
    import liteattention_amd as L
    B, S, H = 1, 2048, 4
    base = [x.cuda() for x in structured_qkv(B, S, H, 128, seed=77)]

    def qkv_at(t):
        g = torch.Generator(device="cuda").manual_seed(1000 + t)
        noise = [0.05 * torch.randn(x.shape, device="cuda", generator=g) for x in base[:2]]
        return (base[0].float() + noise[0]).bfloat16(), (base[1].float() + noise[1]).bfloat16(), base[2]

    thr, trace = L.calibrate_threshold(qkv_at, n_steps=4, target_skip=0.25, tol=0.03)
    assert thr < 0
    assert abs(trace[-2] - 0.25) <= 0.06, trace
    assert trace == sorted(trace)                             # skip fraction is monotone over steps


def test_dynamo_traces_through_the_op_with_its_meta_kernel():
    """torch.compile(fullgraph=True) over a function that calls flash_attn_func: the registered Meta kernel gives dynamo the output
    shapes, the op stays opaque. (AOT-autograd backends reject the op for the output alias annotation `Tensor(out!)` in its schema -
    the reference's own schema, flash_api.cpp:1723-1762, kept verbatim.)"""
    import liteattention_amd as L
    q, k, v = [torch.randn(1, 300, 2, 128, device="cuda").bfloat16() for _ in range(3)]

    def fn(q, k, v):
        return L.flash_attn_func(q * 1.0, k, v).float() * 2.0

    ref = fn(q, k, v)
    out = torch.compile(fn, backend="eager", fullgraph=True)(q, k, v)
    assert torch.equal(out, ref)
