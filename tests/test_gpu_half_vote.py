"""GPU: LA_FLAG_HALF_VOTE - the hand-scheduled head_dim-128 kernel with its skip lists kept per 128-ROW HALF of the 256-row workgroup
(round 6; liteattention_amd/csrc/gen_fwd_x64.py LA_X64_FORM=half, la_fwd_kernel_x64.hip `half_build_walk_wave`). The workgroup walks the
union of its two halves' lists and a wave sits out the tiles only the other half lists; every half must behave exactly like an independent
128-row q-tile walking its own list, i.e. like the oracle at block_m = 128 (the reference's own q-granularity for bf16 head_dim 128,
hopper/_internal/cpp/tile_size.h:35-39): outputs and LSE within the oracle tolerance, write lists bit-exact (1e-3 margin rule of
tests/test_gpu_parity.py). The in-process tests below build the cases where the two halves DIFFER (that is what the form adds); the
subprocess tests run the existing parity files with LA_VOTE=half, so that the same suite holds for both list geometries."""
import math
import os
import subprocess
import sys

import pytest
import torch

from helpers import fragmented_qkv, structured_qkv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BM, BN = 128, 64


@pytest.fixture()
def half(monkeypatch):
    monkeypatch.setenv("LA_VOTE", "half")
    import liteattention_amd as L
    assert L.get_tile_sizes(128, 2) == (128, 64)
    from oracle import oracle as orc
    return L, orc


def _tol(o_ref):
    return 2.0 ** -8 * o_ref.abs().max().item() + 1e-3


def _bad_rows(rd, wr, wr_orc, margins, thr):
    bad = 0
    B, H, Qt = wr_orc.shape[:3]
    for b in range(B):
        for h in range(H):
            for m in range(Qt):
                a, e = wr[b, h, m], wr_orc[b, h, m]
                n = int(e[0])
                if int(a[0]) == n and torch.equal(a[: n + 1], e[: n + 1]):
                    continue
                mg = margins[b, h, m]
                mg = mg[~torch.isnan(mg)]
                if not ((mg - thr).abs() < 1e-3).any():
                    bad += 1
    return bad


def _check_call(L, orc, q, k, v, rd, thr, must_do=None, dtype=torch.bfloat16):
    """One launch on read list `rd` (CPU int32 [B, H, Qt, Kt + 1]) against the oracle at block_m = 128 (which repeats K/V heads itself)."""
    B, Sq, H, _ = q.shape
    Kt = -(-k.shape[1] // BN)
    wr_d = torch.full_like(rd, -7).cuda()
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), attn_read_list=rd.cuda(), attn_write_list=wr_d,
                                 attn_must_do_list=None if must_do is None else must_do.cuda(), thr=thr, return_softmax_lse=True)
    wr_orc = torch.zeros_like(rd)
    margins = torch.empty(B, H, rd.shape[2], Kt)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc, thr=thr, margins=margins,
                                       must_do_list=must_do, p_round="f16" if dtype == torch.float16 else True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    wr = wr_d.cpu()
    assert _bad_rows(rd, wr, wr_orc, margins, thr) == 0
    return wr, wr_orc


def _rows_to_lists(rows, Kt):
    """rows[b][h][m] = python list [start0, end0, ...] -> int32 [B, H, Qt, Kt + 1]."""
    B, H, Qt = len(rows), len(rows[0]), len(rows[0][0])
    out = torch.zeros(B, H, Qt, Kt + 1, dtype=torch.int32)
    for b in range(B):
        for h in range(H):
            for m in range(Qt):
                r = rows[b][h][m]
                out[b, h, m, 0] = len(r)
                out[b, h, m, 1: 1 + len(r)] = torch.tensor(r, dtype=torch.int32)
    return out


@pytest.mark.parametrize("D", [128, 64, 96])
@pytest.mark.parametrize("S", [1000, 1536, 700])
@pytest.mark.parametrize("thr", [-2.0, -5.0])
def test_lists_over_steps_match_the_oracle_at_128_rows(half, S, thr, D):
    """Five denoising-like steps through LiteAttention: lists [2, B, H, ceil(S / 128), Kt + 1]; at every step the oracle gets the kernel's
    read list. S = 1000: the last workgroup's second half is partial (rows 896..999); S = 700: 6 list rows, the last one of 60 rows; S = 1536:
    whole workgroups."""
    L, orc = half
    B, H = 1, 3
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    Qt, Kt = -(-S // BM), -(-S // BN)
    dropped = 0
    for step in range(5):
        q, k, v = fragmented_qkv(B, S, H, D, seed=4, step=step)
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        assert att._skip_list.shape == (2, B, H, Qt, Kt + 1)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        margins = torch.empty(B, H, Qt, Kt)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc, thr=thr, margins=margins,
                                           must_do_list=orc.expand_must_do_ref([0, 0], BN, Kt + 1))
        assert (out.float().cpu() - o_ref).abs().max().item() <= 2.0 ** -7 * o_ref.abs().max().item() + 1e-3, step   # peaked rows: tests/test_gpu_fragmented.py
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3, step
        assert _bad_rows(rd, wr, wr_orc, margins, thr) == 0, step
        dropped = Qt * Kt * H - orc.listed_tiles(wr)
    assert dropped > 0                                 # the thresholds really skip on this data
    # and the two halves of a workgroup really differ somewhere (otherwise this test says nothing about the union walk)
    fin = att.current_read_list().cpu()
    assert any(not torch.equal(fin[0, h, 2 * m], fin[0, h, 2 * m + 1]) for h in range(H) for m in range(Qt // 2))


@pytest.mark.parametrize("D", [128, 64, 96])
def test_halves_with_different_first_tiles_ranges_and_lengths(half, D):
    """Imposed read lists in which the halves of a workgroup share nothing but the kernel: different first tiles (a half that is NOT active
    at the first union position starts from the empty state and masks nothing), ranges that interleave, a half listing one tile, a
    half listing everything, range ends of one half in the middle of the other's ranges; thr = -3 so that write lists are non-trivial."""
    L, orc = half
    B, S, H = 1, 1024 + 77, 2                          # ragged last key tile (Kt = 18: tile 17 holds 13 keys)
    Kt, Qt = -(-S // BN), -(-S // BM)
    q, k, v = structured_qkv(B, S, H, D, seed=21)
    assert Qt == 9 and Kt == 18
    even = [17, 12, 9, 9, 6, 2]
    odd = [14, 13, 11, 10, 8, 7, 5, 5, 1, 0]
    rows = [[[even, odd, [17, 0], [3, 3], [16, 16], [17, 17, 15, 0], odd, even, [17, 5]] for _ in range(H)]]
    rows[0][1] = [[10, 10], [17, 0], odd, [17, 16, 1, 0], [8, 2], even, [0, 0], [17, 0], [9, 8]]
    rd = _rows_to_lists(rows, Kt)
    for thr in (-3.0, -30.0, float("inf")):
        wr, wr_orc = _check_call(L, orc, q, k, v, rd, thr)
        if thr == float("inf"):                       # K1 of the reference's script: every tile flagged -> [2, first, second] of each row's own walk
            assert torch.equal(wr[0, 0, 2, :3], torch.tensor([2, 17, 16], dtype=torch.int32))
            assert torch.equal(wr[0, 0, 1, :3], torch.tensor([2, 14, 13], dtype=torch.int32))
        if thr == -30.0:                              # nothing flagged: write == read
            for h in range(H):
                for m in range(Qt):
                    n = int(rd[0, h, m, 0])
                    assert torch.equal(wr[0, h, m, : n + 1], rd[0, h, m, : n + 1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_must_do_lists_gqa_batch_and_fp16(half, dtype):
    """Batch 2, GQA 4:2, a 4-D must-do list with two ranges (the serial writer under a live mask), fp16 body."""
    L, orc = half
    B, S, H, Hk = 2, 900, 4, 2
    Kt, Qt = -(-S // BN), -(-S // BM)
    q, _, _ = structured_qkv(B, S, H, 128, seed=5, dtype=dtype)
    _, k, v = structured_qkv(B, S, Hk, 128, seed=6, dtype=dtype)
    md_row = orc.expand_must_do_ref([700, 600, 300, 100], BN, Kt + 1)
    md = md_row.view(1, 1, 1, -1).expand(B, H, Qt, Kt + 1).contiguous()
    rd = orc.init_skip_list_ref(B, Qt, Kt, H)[0]
    for step in range(3):
        wr, _ = _check_call(L, orc, q, k, v, rd, -2.5, must_do=md, dtype=dtype)
        rd = wr


def test_long_walk_crosses_many_activity_words(half):
    """Kt = 150 (five 32-position words): the body refills its activity window from LDS every 32 positions; halves alternate in blocks
    whose edges sit on, one before and one after the word boundaries."""
    L, orc = half
    B, S, H = 1, 150 * 64 - 5, 1
    Kt = 150
    g = torch.Generator().manual_seed(8)
    q = torch.randn(B, 256, H, 128, generator=g).bfloat16()
    k = torch.randn(B, S, H, 128, generator=g).bfloat16()
    v = torch.randn(B, S, H, 128, generator=g).bfloat16()
    a = [149, 129, 127, 97, 95, 95, 64, 63, 33, 31, 30, 30, 28, 0]
    b = [148, 130, 128, 128, 96, 96, 94, 65, 62, 32, 29, 29]
    rd = _rows_to_lists([[[a, b]]], Kt)
    _check_call(L, orc, q, k, v, rd, -40.0)
    rd2 = _rows_to_lists([[[b, a]]], Kt)
    _check_call(L, orc, q, k, v, rd2, -1.0)


def test_q_tile_windows_in_pairs_of_halves(half):
    """q-tile windows count 128-row q-tiles: a window starts on an even one and holds an even number unless it reaches the end."""
    L, orc = half
    from liteattention_amd.flash_attn_interface import mha_fwd
    B, S, H = 1, 1400, 2
    Qt, Kt = -(-S // BM), -(-S // BN)                # 11, 22
    q, k, v = structured_qkv(B, S, H, 128, seed=9)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    lists = orc.init_skip_list_ref(B, Qt, Kt, H)
    whole_w = lists[1].clone().cuda()
    o_whole, lse_whole, *_ = mha_fwd(qd, kd, vd, attn_read_list=lists[0].cuda(), attn_write_list=whole_w, thr=-2.0)
    win_w = lists[1].clone().cuda()
    o_win, lse_win, *_ = mha_fwd(qd, kd, vd, attn_read_list=lists[0].cuda(), attn_write_list=win_w, thr=-2.0, _q_windows=[(0, 4), (4, 2), (6, 5)])
    assert torch.equal(o_whole, o_win) and torch.equal(lse_whole, lse_win) and torch.equal(whole_w, win_w)
    for bad in ([(1, 2)], [(0, 3)]):
        with pytest.raises(RuntimeError):
            mha_fwd(qd, kd, vd, attn_read_list=lists[0].cuda(), attn_write_list=win_w, thr=-2.0, _q_windows=bad)


def test_dense_launches_and_other_head_dims_ignore_the_flag(half):
    L, orc = half
    assert L.get_tile_sizes(64, 2) == (128, 64) and L.get_tile_sizes(96, 2) == (128, 64)          # the 256-row kernels all have the form
    assert L.get_tile_sizes(256, 2) == (128, 64) and L.get_tile_sizes(192, 2) == (128, 64) and L.get_tile_sizes(128, 1) == (256, 64)
    q, k, v = structured_qkv(1, 700, 2, 128, seed=3)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=BM, block_n=BN)
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref) and (lse.cpu() - lse_ref).abs().max().item() <= 1e-3


def _suite(args, timeout=1500):
    env = dict(os.environ, LA_VOTE="half")
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("files", [["tests/test_gpu_parity.py", "tests/test_gpu_headline.py"],
                                   ["tests/test_gpu_fragmented.py", "tests/test_gpu_fp16.py", "tests/test_gpu_gqa_windows.py"],
                                   ["tests/test_gpu_varlen_lists.py", "tests/test_gpu_denoise_lists.py", "tests/test_gpu_round2.py"],
                                   ["tests/test_gpu_head_dims.py", "tests/test_gpu_round4.py"]],
                         ids=["parity+headline", "fragmented+fp16+gqa_windows", "varlen+denoise+round2"] + ["head_dims+round4"])
def test_the_parity_suite_under_the_half_vote_geometry(files):
    # (e4m3 and head dims 192 / 256 have no half-vote form - the flag does nothing there and the default run covers them: left out)
    r = _suite(files + ["-k", "not fp8 and not e4m3 and not 192 and not 256"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
