"""GPU: the native fp8 (e4m3) bodies at head dims 64, 96, 192 and 256 (round 6; gen_fwd_x64_fp8.py under LA_X64F8_D, la_fwd_kernel_x64_fp8.hip).

Until round 5 e4m3 at head dims <= 64 ran zero-padded on the head_dim-128 body, and above 128 on the bf16 kernels over up-converted operands.
The native 64 body does one 64-wide contraction per score block and two d-blocks of O^T; the 192 / 256 bodies hold ONE 32-row q-block per wave
(q-tile 128) with 3 / 4 contraction steps and 6 / 8 d-blocks. Every fp32 operation on a real column is the one the next instantiated size does on
zero-padded operands (a zero product adds exactly 0 to a score; d-blocks of O^T are independent), so 64 and 96 must agree with the padded 128 form and 192
with the padded 256 form BIT FOR BIT - O, LSE and the written lists, in all three forms of P. Those are the first tests; the others hold the bodies
against the oracle directly (ragged shapes, lists over steps with descales and GQA, the lazy-rescale path, head dims served by padding onto these
bodies) as tests/test_gpu_fp8.py does at 128; packed variable-length batches: tests/test_gpu_varlen_lists.py. The reference-generated golden
`fp8_sq130_sk517_h2_d64` runs in test_gpu_fp8.py::test_fp8_dense_matches_reference_outputs."""
import pytest
import torch

from helpers import fp8_lse_tol, fp8_p_round, structured_qkv

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn
DIMS = [64, 96, 192, 256]
PADDED = {64: 128, 96: 128, 192: 256}   # head dim -> a larger instantiated size its zero-padded form runs on


@pytest.fixture(params=["encoded", "exp", "exact"], autouse=True)
def p_mode(request, monkeypatch):
    """The three forms of P (tests/test_gpu_fp8.py)."""
    monkeypatch.delenv("LA_FP8_P", raising=False)
    if request.param == "encoded":
        monkeypatch.setenv("LA_FP8_P", "encoded")
    elif request.param == "exp":
        monkeypatch.setenv("LA_FP8_P", "mfma_rowsum")
    return request.param


def _tiles(D):
    import liteattention_amd as L
    return L.get_tile_sizes(D, 1)


def _tol(o):
    return 0.05 * o.abs().max().item() + 2e-2


def _pad(t, to):
    return torch.nn.functional.pad(t.view(torch.uint8), (0, to - t.shape[-1])).view(F8)


def test_the_library_serves_these_head_dims_natively():
    import liteattention_amd as L
    from liteattention_amd import _cabi
    from liteattention_amd.flash_attn_interface import kernel_head_dim
    assert all(_cabi.is_instantiated(d, 1, 0) for d in (64, 96, 128, 192, 256))
    assert [kernel_head_dim(d, 1) for d in (48, 64, 80, 96, 112, 128, 144, 192, 208, 256)] == [64, 64, 96, 96, 128, 128, 192, 192, 256, 256]
    assert L.get_tile_sizes(64, 1) == (256, 64) and L.get_tile_sizes(192, 1) == L.get_tile_sizes(256, 1) == (128, 64)


@pytest.mark.parametrize("D", sorted(PADDED))
@pytest.mark.parametrize("shape", [(2, 300, 4, 2, 1000), (1, 17, 1, 1, 17), (1, 700, 2, 2, 4224)])
def test_native_body_equals_the_zero_padded_next_size_bit_for_bit_dense(shape, D):
    import liteattention_amd as L
    B, Sq, H, Hk, Sk = shape
    g = torch.Generator().manual_seed(Sq + Sk)
    q, k, v = (torch.randn(B, Sq, H, D, generator=g).to(F8).cuda(), torch.randn(B, Sk, Hk, D, generator=g).to(F8).cuda(),
               torch.randn(B, Sk, Hk, D, generator=g).to(F8).cuda())
    qd, kd, vd = [(0.5 + torch.rand(B, Hk, generator=g)).cuda() for _ in range(3)]
    out, lse = L.flash_attn_func(q, k, v, q_descale=qd, k_descale=kd, v_descale=vd, return_softmax_lse=True)
    P = PADDED[D]                      # (padding to 128 makes 96 run on the 128 body: the library never pads 96 itself any more)
    out_p, lse_p = L.flash_attn_func(_pad(q, P), _pad(k, P), _pad(v, P), softmax_scale=D ** -0.5, q_descale=qd, k_descale=kd, v_descale=vd,
                                     return_softmax_lse=True)
    assert out.shape == (B, Sq, H, D) and bool(torch.isfinite(out.float()).all())
    assert torch.equal(out, out_p[..., :D]) and torch.equal(lse, lse_p)
    assert not out_p[..., D:].any()


@pytest.mark.parametrize("D", sorted(PADDED))
def test_native_body_equals_the_zero_padded_next_size_bit_for_bit_lists(D):
    """Three steps of lists on two LiteAttention objects (native / padded inputs): O, LSE and both lists equal after every step."""
    import liteattention_amd as L
    B, S, H, thr, P = 1, 2304, 3, -3.0, PADDED[D]
    a64, a128 = L.LiteAttention(threshold=thr, max_batch_size=B), L.LiteAttention(threshold=thr, max_batch_size=B)
    listed = []
    for step in range(3):
        q, k, v = [x.to(F8).cuda() for x in structured_qkv(B, S, H, D, seed=640 + step, alpha=9.0, dtype=torch.float32)]
        out, lse = a64(q, k, v, return_softmax_lse=True)
        out_p, lse_p = a128(_pad(q, P), _pad(k, P), _pad(v, P), scale=D ** -0.5, return_softmax_lse=True)
        assert torch.equal(out, out_p[..., :D]) and torch.equal(lse, lse_p)
        for i in (0, 1):
            assert torch.equal(a64._skip_list[i], a128._skip_list[i])
        from oracle import oracle as orc
        listed.append(orc.listed_tiles(a64._skip_list[a64._phase][:B].cpu()))
    Qt, Kt = -(-S // _tiles(D)[0]), -(-S // _tiles(D)[1])
    assert listed[-1] < 0.95 * B * H * Qt * Kt


@pytest.mark.parametrize("shape", [(1, 17, 1, 17, 64), (2, 129, 3, 65, 64), (1, 1000, 2, 1250, 64), (1, 128, 1, 4224, 64), (1, 300, 2, 700, 48),
                                   (1, 260, 2, 130, 32), (1, 70, 1, 333, 16), (2, 129, 3, 65, 96), (1, 1000, 2, 1250, 96), (1, 300, 2, 700, 80), (1, 17, 1, 17, 256), (2, 129, 3, 65, 192), (1, 1000, 2, 1250, 256),
                                   (1, 128, 1, 4224, 192), (1, 300, 2, 700, 160), (1, 260, 2, 130, 224)])
def test_ragged_shapes_against_the_oracle(shape):
    import liteattention_amd as L
    from oracle import oracle as orc
    B, Sq, H, Sk, d = shape
    bm, bn = L.get_tile_sizes(d, 1)
    g = torch.Generator().manual_seed(Sq * 7 + Sk + d)
    q, k, v = [torch.randn(B, s, H, d, generator=g).to(F8) for s in (Sq, Sk, Sk)]
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    o8, lse8, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=fp8_p_round())
    assert out.shape == q.shape
    assert (out.float().cpu() - o8).abs().max().item() <= _tol(o8)
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()


@pytest.mark.parametrize("D", DIMS)
def test_skip_lists_match_the_oracle_over_steps_with_descales_and_gqa(D):
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    bm, bn = _tiles(D)
    B, S, H, Hk, thr = 1, 2560, 4, 2, -3.0           # 40 key tiles: the vote words wrap (32 bits per word)
    Qt, Kt = S // bm, S // bn
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    md_row = orc.expand_must_do_ref([0, 0], bn, Kt + 1)
    margins = torch.empty(B, H, Qt, Kt)
    qd, kd, vd = torch.tensor([[0.7, 1.3]]), torch.tensor([[1.1, 0.9]]), torch.tensor([[0.5, 1.7]])
    listed = []
    for step in range(4):
        q, k, v = [x.to(F8) for x in structured_qkv(B, S, H, D, seed=364, alpha=9.0, dtype=torch.float32)]
        k, v = k[:, :, :Hk].contiguous(), v[:, :, :Hk].contiguous()
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True, q_descale=qd.cuda(), k_descale=kd.cuda(), v_descale=vd.cuda())
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, must_do_list=md_row, thr=thr,
                                           margins=margins, p_round=fp8_p_round(), q_descale=qd, k_descale=kd, v_descale=vd)
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, thr, B)
        assert bad == 0
        listed.append(orc.listed_tiles(wr[:B]))
    assert listed[-1] < 0.95 * B * H * Qt * Kt and listed == sorted(listed, reverse=True)


@pytest.mark.parametrize("D", DIMS)
def test_running_max_that_grows_late_in_the_walk(D):
    """The O^T rescale round trip of these bodies (2 / 6 / 8 d-blocks per q-block): tests/test_gpu_fp8.py, gain 8."""
    import liteattention_amd as L
    from oracle import oracle as orc
    from test_gpu_parity import _compare_lists
    bm, bn = _tiles(D)
    B, S, H = 1, 1536, 2
    g = torch.Generator().manual_seed(93)
    q, k, v = [torch.randn(B, S, H, D, generator=g) for _ in range(3)]
    k = k * torch.linspace(8.0, 1.0, S).view(1, S, 1, 1)
    q, k, v = [x.to(F8) for x in (q, k, v)]
    o8, lse8, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=fp8_p_round())
    out, lse = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
    assert bool(torch.isfinite(out.float()).all())
    assert (out.float().cpu() - o8).abs().max().item() <= _tol(o8)
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()
    Qt, Kt = -(-S // bm), -(-S // bn)
    att = L.LiteAttention(threshold=-1.0, max_batch_size=B)
    margins = torch.empty(B, H, Qt, Kt)
    for _ in range(2):
        rd_idx = att._phase if att._skip_list is not None else 0
        out, lse = att(q.cuda(), k.cuda(), v.cuda(), return_softmax_lse=True)
        rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
        wr_orc = torch.zeros_like(wr)
        o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, read_list=rd, write_list=wr_orc, thr=-1.0, margins=margins,
                                           p_round=fp8_p_round())
        assert (out.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (lse.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()
        bad, _ = _compare_lists(orc, rd, wr, wr_orc, margins, -1.0, B)
        assert bad == 0


@pytest.mark.parametrize("D", DIMS)
def test_static_map_equals_the_ticket_queues_and_q_windows_compose(D):
    """Raw mha_fwd: the static one-workgroup-per-item map and two q-tile windows give the results of the default launch."""
    from liteattention_amd.flash_attn_interface import mha_fwd
    g = torch.Generator().manual_seed(5)
    B, S, H = 2, 1100, 3
    q, k, v = [torch.randn(B, S, H, D, generator=g).to(F8).cuda() for _ in range(3)]
    ref = mha_fwd(q, k, v)
    st = mha_fwd(q, k, v, _static_sched=True)
    assert torch.equal(ref[0], st[0]) and torch.equal(ref[1], st[1])
    Qt = -(-S // _tiles(D)[0])
    win = mha_fwd(q, k, v, _q_windows=[(0, 2), (2, Qt - 2)])
    assert torch.equal(ref[0], win[0]) and torch.equal(ref[1], win[1])


@pytest.mark.parametrize("D", [64, 256])
def test_dense_key_range_beyond_one_launchs_walk(D, p_mode):
    """The walk of a launch lives in LDS (~6 400 key tiles at head_dim <= 128, ~4 800 at 192 / 256 for e4m3). A DENSE call beyond that is cut into
    runs inside la_fwd (per run: V^T prepare pass of its keys, forward on partial O / LSE, then the LSE merge), as for bf16 / fp16
    (tests/test_gpu_head_dims.py); a call with skip lists keeps the bound and its typed error. Against fp32 torch on the same e4m3 inputs."""
    import liteattention_amd as L
    from liteattention_amd import _cabi
    g = torch.Generator(device="cuda").manual_seed(D)
    Sk = (6600 if D == 64 else 5000) * 64 - 9
    q = torch.randn(1, 131, 1, D, device="cuda", generator=g).to(F8)
    k = torch.randn(1, Sk, 1, D, device="cuda", generator=g).to(F8)
    v = torch.randn(1, Sk, 1, D, device="cuda", generator=g).to(F8)
    out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    sc = q.float()[0, :, 0] @ k.float()[0, :, 0].T / D ** 0.5
    ref = torch.softmax(sc, -1) @ v.float()[0, :, 0]
    assert bool(torch.isfinite(out.float()).all())
    assert (out.float()[0, :, 0] - ref).abs().max().item() <= 0.05 * ref.abs().max().item() + 2e-3
    assert (lse[0, 0] - torch.logsumexp(sc, -1)).abs().max().item() <= (1e-3 if p_mode == "exact" else 5e-3)
    bm, bn = L.get_tile_sizes(D, 1)
    lists = torch.zeros(2, 1, 1, -(-131 // bm), -(-Sk // bn) + 1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match=_cabi.status_string(_cabi.LA_ERR_SEQLEN)[:20]):
        L.flash_attn_func(q, k, v, attn_read_list=lists[0], attn_write_list=lists[1])


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("B", [1, 2])
def test_host_split_kv_of_e4m3_calls(D, B):
    """`num_splits` for e4m3 (round 6: batch 1 takes the one-sequence path - the splits are the batch of ONE fixed-length launch, q and the
    (1, Hk) descales with batch stride 0, bf16 partials merged by la_combine; batch > 1 the packed path). Every split rounds P to e4m3 against
    ITS running maximum, so O agrees with the unsplit launch inside the fp8 bound, not to a bf16 step; the LSE (fp32 row sums of the
    un-rounded P in the default form) agrees to 1e-5 - by form of P otherwise - and with the oracle."""
    import liteattention_amd as L
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(40 + D + B)
    Sq, Sk, H, Hk = 300, 9000, 4, 2
    q, k, v = torch.randn(B, Sq, H, D, generator=g).to(F8), torch.randn(B, Sk, Hk, D, generator=g).to(F8), torch.randn(B, Sk, Hk, D, generator=g).to(F8)
    qd, kd, vd = [0.5 + torch.rand(B, Hk, generator=g) for _ in range(3)]
    kw = dict(q_descale=qd.cuda(), k_descale=kd.cuda(), v_descale=vd.cuda(), return_softmax_lse=True)
    o1, l1 = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), **kw)
    bm, bn = L.get_tile_sizes(D, 1)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round=fp8_p_round(), q_descale=qd, k_descale=kd, v_descale=vd)
    for ns in (3, 4):
        o2, l2 = L.flash_attn_func(q.cuda(), k.cuda(), v.cuda(), num_splits=ns, **kw)
        assert o2.dtype == torch.bfloat16 and o2.shape == o1.shape
        assert (o2.float() - o1.float()).abs().max().item() <= _tol(o_ref)
        assert (o2.float().cpu() - o_ref).abs().max().item() <= _tol(o_ref)
        assert (l2 - l1).abs().max().item() <= fp8_lse_tol() and (l2.cpu() - lse_ref).abs().max().item() <= fp8_lse_tol()


@pytest.mark.parametrize("D", DIMS)
def test_race_screen_100_iterations(D):
    """The reference's race screen (hopper/tests/test_flash_attn.py:1144-1175; tests/test_gpu_head_dims.py for the bf16 bodies) on the e4m3 bodies
    of this module: real, fragmented, unequal lists, more items than CUs (persistent loop, ticket stealing), a co-running memory hog on a
    second stream; every launch identical to the first in O, LSE and the write list (the V^T prepare pass rewrites the same workspace bytes)."""
    import liteattention_amd as L
    B, S, H, thr = 1, 8192, 8, -2.0
    q, k, v = [x.to(F8).cuda() for x in structured_qkv(B, S, H, D, seed=700, alpha=7.0, frames=16, dtype=torch.float32)]
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
    for it in range(100):
        if it % 4 == 0:
            with torch.cuda.stream(hog_stream):
                hog_b.copy_(hog_a)
        out, lse = run()
        assert torch.equal(out, out0) and torch.equal(lse, lse0) and torch.equal(att._skip_list, lists0), it
    torch.cuda.synchronize()
