"""GPU: the REAL lists of BASELINE.json configs[2] — 50 synthetic denoising steps at B=1, S=75 600, H=40, D=128 through
``LiteAttention.__call__`` at the two thresholds of the committed runs (thr -4.22 -> ~44 %, -2.46 -> ~78 % last-step sparsity) —
checked against references at the size they run at (VERDICT r2 "what's weak" 1b; the mid-size oracle counterpart is
tests/test_gpu_fragmented.py). At this size the CPU oracle cannot finish, so the checkers are fp32 torch restatements on the
device (tools/selfcheck.py), each citing the reference lines it follows:

  * step-49 output: 256 sampled rows x 3 heads vs fp32 attention over exactly the keys the row's q-tile READ list names
        bf16 |O - ref| <= 2^-7 max|ref| + 1e-4 (one bf16 ulp at the maximum: these rows are peaked — a few keys carry 5-10 % of
        the weight each — and under the lazy rescale the bf16 rounding of a dominant P does not vanish as it does for P = 2^0:
        tests/test_gpu_fragmented.py docstring; measured 0.0031 at max|ref| 0.52), |LSE - ref| <= 2e-4;
        fp8 <= 0.05 max|ref| + 1e-3, |LSE - ref| <= 2e-2 (row sums of the encoded P, helpers.fp8_lse_tol)
  * step-49 write list: for 24 sampled (head, q-tile) rows the skip vote of every walked tile (softmax.h:190-194) and the
    writer state machine (mainloop...:142-192) restated in torch; rows equal except those with a tile within 1e-3 of thr
  * walked(write) is a subset of walked(read) for ALL 11 840 rows (a skipped tile is never revisited), both start at Kt-1
  * dynamic work distribution == static map bit-exactly (O, LSE, write list) on these lists
  * the lists really are fragmented: some row holds more than 64 ranges
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

S, H, D = 75600, 40, 128
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module", params=[-4.22, -2.462], ids=["thr-4.22", "thr-2.46"])
def run49(request):
    """50 steps; keeps what step 49 read, wrote and returned."""
    import liteattention_amd as L
    from tools.selfcheck import DenoiseWorkload
    thr = request.param
    wl = DenoiseWorkload(H, torch.device("cuda", 0))
    att = L.LiteAttention(threshold=thr, max_batch_size=1)
    for t in range(wl.steps - 1):
        q, k, v = wl.qkv(t)
        att(q, k, v)
        del q, k, v
    q, k, v = wl.qkv(wl.steps - 1)
    read = att.current_read_list().clone()
    att._skip_list[1 - att._phase].fill_(-7)            # the kernel must write every live entry of the write list
    out, lse = att(q, k, v, return_softmax_lse=True)
    write = att.current_read_list().clone()
    del wl
    return dict(thr=thr, q=q, k=k, v=v, read=read, write=write, out=out, lse=lse)


def _items(qt, n=24, seed=3):
    g = torch.Generator().manual_seed(seed)
    hs = torch.randint(0, H, (n,), generator=g).tolist()
    ms = torch.randint(0, qt, (n,), generator=g).tolist()
    ms[0], ms[1], ms[2] = 0, qt - 1, qt - 2             # first q-tile, the zero-padded last one, its neighbour
    return list(zip(hs, ms))


def test_step49_bf16(run49):
    import liteattention_amd as L
    from tools import selfcheck as sc
    from liteattention_amd.flash_attn_interface import mha_fwd
    r = run49
    bm, bn = L.get_tile_sizes(D, 2)
    qt, kt = -(-S // bm), -(-S // bn)
    read, write = r["read"], r["write"]
    assert torch.equal(read[..., 1], torch.full_like(read[..., 1], kt - 1)) and torch.equal(write[..., 1], read[..., 1])
    assert int(read[..., 0].max()) > 128, "no row with more than 64 ranges: these lists are not fragmented"
    frac = sc.lists_to_bitmap(read).float().mean().item()
    if bm == 256:
        assert (0.50 < frac < 0.62) if r["thr"] < -4 else (0.18 < frac < 0.28), frac      # ~44 % / ~78 % sparsity (profiles/r01e)
    else:          # the 128-row vote (LA_VOTE=half / LA_FWD_KERNEL=v2) drops more at a threshold: 52.5 % / 83.7 % here
        assert (0.42 < frac < 0.54) if r["thr"] < -4 else (0.12 < frac < 0.22), frac
    # a skipped tile is never revisited
    bm_r, bm_w = sc.lists_to_bitmap(read), sc.lists_to_bitmap(write)
    assert int((bm_w & ~bm_r).sum()) == 0
    assert int(bm_w.sum()) < int(bm_r.sum())                                            # and the step still drops tiles
    # output of the step against fp32 torch over the listed keys
    res = sc.sampled_row_check(r["q"], r["k"], r["v"], r["out"], r["lse"], read, bm, bn, heads=(0, 17, 39), n_rows=256,
                               o_rtol=2.0 ** -7)
    assert res["ok"], res
    # vote + writer restated
    vw = sc.vote_writer_check(r["q"], r["k"], read, write, r["thr"], bm, bn, _items(qt))
    assert vw["ok"] and vw["items"] == 24 and vw["borderline"] <= 2, vw
    # static map, NaN-prefilled output: bit-identical to the dynamic run
    out_s = torch.full_like(r["out"], float("nan"))
    wr_s = torch.full_like(read, -7)
    must_do = torch.zeros(kt + 1, dtype=torch.int32, device="cuda")
    must_do[0] = 2
    _, lse_s, *_ = mha_fwd(r["q"], r["k"], r["v"], out=out_s, attn_read_list=read, attn_must_do_list=must_do,
                           attn_write_list=wr_s, thr=r["thr"], _must_do_is_1d=True, _static_sched=True)
    assert torch.equal(out_s, r["out"]) and torch.equal(lse_s, r["lse"])
    n = int(write[..., 0].max().item())
    live = torch.arange(n + 1, device="cuda") <= write[..., 0:1]
    assert bool(((wr_s[..., : n + 1] == write[..., : n + 1]) | ~live).all())


def test_step49_fp8_on_the_same_lists(run49):
    """The fp8 kernel walks the SAME fragmented read lists (same 256 x 64 tile geometry) on the e4m3 cast of step 49's tensors."""
    import liteattention_amd as L
    from tools import selfcheck as sc
    from liteattention_amd.flash_attn_interface import mha_fwd
    r = run49
    if L.get_tile_sizes(D, 1) != L.get_tile_sizes(D, 2):
        pytest.skip("the bf16 lists of this run use another q-tile than the fp8 kernel's (LA_VOTE=half / LA_FWD_KERNEL=v2)")
    bm, bn = L.get_tile_sizes(D, 1)
    qt, kt = -(-S // bm), -(-S // bn)
    q, k, v = [x.to(F8) for x in (r["q"], r["k"], r["v"])]
    read = r["read"]
    must_do = torch.zeros(kt + 1, dtype=torch.int32, device="cuda")
    must_do[0] = 2
    wr = torch.full_like(read, -7)
    out, lse = L.flash_attn_func(q, k, v, attn_read_list=read, attn_must_do_list=must_do, attn_write_list=wr, thr=r["thr"],
                                 return_softmax_lse=True)
    res = sc.sampled_row_check(q, k, v, out, lse, read, bm, bn, heads=(0, 17, 39), n_rows=256, o_rtol=0.05, o_atol=1e-3,
                               lse_atol=2e-2)
    assert res["ok"], res
    vw = sc.vote_writer_check(q, k, read, wr, r["thr"], bm, bn, _items(qt, seed=4))
    assert vw["ok"] and vw["borderline"] <= 2, vw
    assert int((sc.lists_to_bitmap(wr) & ~sc.lists_to_bitmap(read)).sum()) == 0
    out_s = torch.full_like(out, float("nan"))
    wr_s = torch.full_like(read, -7)
    _, lse_s, *_ = mha_fwd(q, k, v, out=out_s, attn_read_list=read, attn_must_do_list=must_do, attn_write_list=wr_s,
                           thr=r["thr"], _must_do_is_1d=True, _static_sched=True)
    assert torch.equal(out_s, out) and torch.equal(lse_s, lse)
    n = int(wr[..., 0].max().item())
    live = torch.arange(n + 1, device="cuda") <= wr[..., 0:1]
    assert bool(((wr_s[..., : n + 1] == wr[..., : n + 1]) | ~live).all())
