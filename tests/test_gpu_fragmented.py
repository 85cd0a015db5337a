"""GPU: threshold-produced, FRAGMENTED skip lists against a reference ("what's weak" 1).

Real QK-Skip lists hold many short ranges per row; the paths they exercise — the multi-pass branch of the wave-parallel
read-list expansion (> 64 ranges per row, la_fwd_common.h `expand_read_list`), the cross-chunk carry of the wave-parallel
writer (`write_skip_list_wave`) and the tile-address table on a walk that jumps every few tiles — must be compared with the
oracle, not just screened for determinism. Mid-size here (S = 16 300: Kt = 255, up to ~75 ranges per row; the oracle takes
seconds); the full-size counterpart on the step-49 lists of the 50-step run is tests/test_gpu_denoise_lists.py.

Reference lines restated by the oracle: mainloop_fwd_sm90_tma_gmma_ws.hpp:47-192 (reader / writer), :1804-1827 (walk).
Tolerances: LSE <= 1e-3 (fp8: helpers.fp8_lse_tol()); write lists bit-exact except rows holding a tile whose decision margin is
within 1e-3 of the threshold; O (fp8: 0.05 max|O| + 2e-2 as in test_gpu_fp8.py):

    |O - oracle| <= 2^-7 max|O| + 1e-3        default kernels (lazy rescale, tau = 8)
    |O - oracle| <= 2^-8 max|O| + 1e-3        LA_FLAG_EXACT_RESCALE (checked on the last step of the bf16 head_dim-128 case)

Why the default bound is one bf16 ulp at max|O| here and half an ulp in test_gpu_parity.py: this generator makes PEAKED rows (a
few keys carry most of a row's weight, max|O| > 1). Both the reference and the kernel round P to bf16 before P V while the row sum
uses the un-rounded P (softmax.h:275-296, mainloop...:1645-1647). With an exact rescale the dominant key of a row has P = 2^0:
no rounding error where it matters. Under the lazy rescale (O kept relative to a reference max that lags the true one by up to
2^8, HISTORY.md 3.1) the dominant P is an arbitrary value in [1, 2^8]: it carries a relative rounding error of up to 2^-9 that
scales the whole row of O. Measured (round 3, head_dim 256, step 2): 0.00795 at |O| = 1.19 lazily, 0.0032 exactly rescaled;
on flat rows (random data, test_gpu_parity.py) the rounding errors of thousands of keys average out and 2^-8 holds either way.
"""
import math

import pytest
import torch

from helpers import fp8_lse_tol, fragmented_qkv, fp8_p_round
from test_gpu_parity import _compare_lists

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def _setup(dtype, D):
    import liteattention_amd as L
    from oracle import oracle as orc
    fp8 = dtype == "fp8"
    bm, bn = L.get_tile_sizes(D, 1 if fp8 else 2)
    if fp8:
        cast, p_round = (lambda x: x.to(F8)), fp8_p_round()
        tol = lambda o: 0.05 * o.abs().max().item() + 2e-2                            # noqa: E731
    elif dtype == "fp16":
        cast, p_round = (lambda x: x.half()), "f16"
        tol = lambda o: 2.0 ** -10 * o.abs().max().item() + 1e-3                      # noqa: E731
    else:
        cast, p_round = (lambda x: x.bfloat16()), True
        tol = lambda o: 2.0 ** -7 * o.abs().max().item() + 1e-3                       # noqa: E731   (module docstring)
    lse_tol = fp8_lse_tol() if fp8 else 1e-3
    return L, orc, bm, bn, cast, p_round, tol, lse_tol


@pytest.mark.parametrize("dtype,D,H", [("bf16", 128, 8), ("fp8", 128, 8), ("bf16", 64, 4), ("bf16", 256, 4), ("fp16", 96, 2)])
def test_fragmented_lists_match_oracle_over_steps(dtype, D, H):
    """6 steps of the fragmenting generator at thr = -3 through LiteAttention.__call__ (dynamic work distribution). Every step
    the oracle walks the SAME read list: O, LSE and the write list must agree; some rows must hold more than 64 ranges."""
    L, orc, bm, bn, cast, p_round, tol, lse_tol = _setup(dtype, D)
    B, S, thr, steps = 1, 16300, -3.0, 6
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    md_row = orc.expand_must_do_ref([0, 0], bn, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    max_len, listed, borderline = 0, [], 0
    for step in range(steps):
        q, k, v = [cast(x) for x in fragmented_qkv(B, S, H, D, seed=5, step=step, steps=steps, dtype=torch.float32)]
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        max_len = max(max_len, int(rd[..., 0].max()))
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc,
                                                 must_do_list=md_row, thr=thr, margins=margins, p_round=p_round)
        assert n_tiles == orc.listed_tiles(rd[:B])
        assert (out.float().cpu() - o_ref).abs().max().item() <= tol(o_ref), f"step {step}"
        assert (lse.cpu() - lse_ref).abs().max().item() <= lse_tol, f"step {step}"
        bad, border = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0, f"step {step}: {bad} rows differ from the oracle with no borderline tile"
        borderline += border
        listed.append(orc.listed_tiles(wr[:B]))
    assert max_len > 128, f"longest read row holds {max_len // 2} ranges: the multi-pass expansion was not exercised"
    if dtype == "bf16" and D in (128, 256):
        # the same read list with an exact rescale (LA_FLAG_EXACT_RESCALE): the half-ulp bound holds, and the lists do not depend on it
        from liteattention_amd import _cabi
        from liteattention_amd.flash_attn_interface import mha_fwd
        wr2 = torch.zeros_like(rd).cuda()
        o2, lse2, *_ = mha_fwd(q.cuda(), k.cuda(), v.cuda(), attn_read_list=rd.cuda(), attn_must_do_list=md_row.cuda(),
                               attn_write_list=wr2, thr=thr, _must_do_is_1d=True, _flags=_cabi.LA_FLAG_EXACT_RESCALE)
        assert (o2.float().cpu() - o_ref).abs().max().item() <= 2.0 ** -8 * o_ref.abs().max().item() + 1e-3
        assert (lse2.cpu() - lse_ref).abs().max().item() <= 1e-3
        n = int(wr[..., 0].max())
        live = torch.arange(n + 1) <= wr[..., 0:1]
        assert bool(((wr2.cpu()[..., : n + 1] == wr[..., : n + 1]) | ~live).all())
    assert listed == sorted(listed, reverse=True) and listed[-1] < 0.7 * B * H * Qt * Kt
    assert borderline <= 4


@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("fp8", 128), ("bf16", 64), ("bf16", 192)])
@pytest.mark.parametrize("thr", [float("-inf"), -2.0])
def test_hand_built_alternating_list_through_flash_attn_func(dtype, D, thr):
    """Keep every other tile: Kt / 2 single-tile ranges per row (the densest fragmentation a list can hold), handed to
    flash_attn_func directly. At thr = -inf the written list names the same tiles; at thr = -2 single flagged tiles are never
    dropped (the first flagged tile of a run survives) — either way O, LSE and the list equal the oracle's."""
    L, orc, bm, bn, cast, p_round, tol, lse_tol = _setup(dtype, D)
    B, S, H = 2, 8300, 3
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    q, k, v = [cast(x) for x in fragmented_qkv(B, S, H, D, seed=9, dtype=torch.float32)]
    lists = torch.zeros(2, B, H, Qt, Kt + 1, dtype=torch.int32)
    kept = list(range(Kt - 1, -1, -2))                                  # Kt-1, Kt-3, ...
    row = [2 * len(kept)] + [t for n in kept for t in (n, n)]
    lists[0, ..., : len(row)] = torch.tensor(row, dtype=torch.int32)
    # odd q-tiles of head 1: ranges of two tiles with one-tile holes (walks 2 of every 3 tiles)
    kept2 = [(n, n - 1) for n in range(Kt - 1, 0, -3)]
    row2 = [2 * len(kept2)] + [t for pr in kept2 for t in pr]
    lists[0, :, 1, 1::2] = 0
    lists[0, :, 1, 1::2, : len(row2)] = torch.tensor(row2, dtype=torch.int32)
    md_row = orc.expand_must_do_ref([0, 0], bn, Kt + 1)
    dl = lists.cuda()
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), attn_read_list=dl[0], attn_write_list=dl[1],
                                 attn_must_do_list=md_row.cuda(), thr=thr, return_softmax_lse=True)
    margins = torch.empty(B, H, Qt, Kt)
    wr_orc = torch.zeros_like(lists[1])
    o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=lists[0], write_list=wr_orc,
                                             must_do_list=md_row, thr=thr, margins=margins, p_round=p_round)
    assert n_tiles == orc.listed_tiles(lists[0])
    assert (out.float().cpu() - o_ref).abs().max().item() <= tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= lse_tol
    wr = dl[1].cpu()
    bad, border = _compare_lists(orc, lists[0], wr, wr_orc, margins, thr, B)
    assert bad == 0 and border <= 2
    assert torch.equal(dl[0].cpu(), lists[0])                           # the read list is never written
    if thr == float("-inf"):
        assert orc.walk_tiles(wr[0, 0, 0].tolist()) == kept
        assert orc.walk_tiles(wr[1, 1, 1].tolist()) == [t for pr in kept2 for t in pr]


def _random_row(kt, g, keep_first=True):
    """A random VALID read-list row: descending inclusive ranges, even length; starts with tile kt - 1 when asked."""
    n_cuts = int(torch.randint(0, max(1, kt // 2), (1,), generator=g))
    cuts = sorted(set(torch.randint(0, kt, (2 * n_cuts,), generator=g).tolist()), reverse=True)
    if len(cuts) % 2:
        cuts = cuts[:-1]
    pairs = [(cuts[i], cuts[i + 1]) for i in range(0, len(cuts), 2)]
    ranges, last_end = [], kt
    for s_, e_ in pairs:                                  # keep them disjoint and strictly descending
        s_ = min(s_, last_end - 1)
        if s_ < 0 or e_ > s_:
            continue
        ranges.append((s_, e_)); last_end = e_
    if keep_first and (not ranges or ranges[0][0] != kt - 1):
        ranges = [(kt - 1, kt - 1)] + [r for r in ranges if r[0] < kt - 1]
    if not ranges:
        ranges = [(kt - 1, 0)]
    return [2 * len(ranges)] + [t for r in ranges for t in r]


@pytest.mark.parametrize("dtype,D", [("bf16", 128), ("fp8", 128), ("bf16", 64), ("bf16", 96), ("bf16", 256), ("fp16", 192)])
@pytest.mark.parametrize("seed", [0, 1])
def test_random_valid_lists_with_must_do_ranges_match_the_oracle(dtype, D, seed):
    """Every (batch, head, q-tile) row gets its OWN random valid list (0 ... Kt / 2 ranges of random lengths; some rows do not start at
    tile Kt - 1, so the ragged last key tile is masked away from the first walked position), a multi-range must-do list keeps the
    serial writer in play (mainloop...:154-162), Sq != Sk, GQA, ragged lengths. One call at thr = -1.5: O, LSE and the write list
    against the oracle on the same list. Complements the generator-driven tests above: nothing here is structured."""
    L, orc, bm, bn, cast, p_round, tol, lse_tol = _setup(dtype, D)
    B, Sq, Sk, H, Hk, thr = 2, 1100, 4000 - 13, 4, 2, -1.5
    Qt, Kt = math.ceil(Sq / bm), math.ceil(Sk / bn)
    g = torch.Generator().manual_seed(100 + seed)
    u = torch.randn(B, 1, Hk, D, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    gain = torch.linspace(0.1, 1.0, Sk).view(1, Sk, 1, 1)                         # late keys (walked first) score highest: flags fire further down
    q = cast(9.0 * u.repeat_interleave(H // Hk, dim=2) + 0.5 * torch.randn(B, Sq, H, D, generator=g))
    k = cast(9.0 * gain * u + 0.5 * torch.randn(B, Sk, Hk, D, generator=g))
    v = cast(torch.randn(B, Sk, Hk, D, generator=g))
    lists = torch.zeros(2, B, H, Qt, Kt + 1, dtype=torch.int32)
    for b in range(B):
        for h in range(H):
            for m in range(Qt):
                row = _random_row(Kt, g, keep_first=(b + h + m) % 4 != 0)
                lists[0, b, h, m, : len(row)] = torch.tensor(row, dtype=torch.int32)
    must_do_tokens = [Sk - 200, Sk - 900, 1500, 1100, 300, 0]                        # three ranges, descending (README.md:180-191)
    md_row = orc.expand_must_do_ref(must_do_tokens, bn, Kt + 1)
    dl = lists.cuda()
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), attn_read_list=dl[0], attn_write_list=dl[1],
                                 attn_must_do_list=md_row.cuda(), thr=thr, return_softmax_lse=True)
    margins = torch.empty(B, H, Qt, Kt)
    wr_orc = torch.zeros_like(lists[1])
    o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=lists[0], write_list=wr_orc,
                                             must_do_list=md_row, thr=thr, margins=margins, p_round=p_round)
    assert n_tiles == orc.listed_tiles(lists[0])
    assert (out.float().cpu() - o_ref).abs().max().item() <= tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= lse_tol
    wr = dl[1].cpu()
    bad, border = _compare_lists(orc, lists[0], wr, wr_orc, margins, thr, B)
    assert bad == 0 and border <= 3
    assert torch.equal(dl[0].cpu(), lists[0])
    assert orc.listed_tiles(wr) < orc.listed_tiles(lists[0])                          # and the call really dropped tiles
