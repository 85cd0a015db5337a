"""GPU: the reference's text + video recipe END TO END (/root/reference/README.md:225-246): full self-attention over [text; video] tokens
broken into t2t, t2v, v2t - dense, ``enable_skip_optimization(False)`` - and v2v with QK-Skip, on ONE ``LiteAttention`` object, the partial
results merged by their LSE. The README leaves the merge to the caller; here it is ``flash_attn_combine`` on the separate partials
(C-ABI ``la_combine_list``: no stacking copy). Checked: the merged result against the oracle's parts merged by ``attention_combine_ref``
(hopper/tests/test_flash_attn.py:1178-1187) and, at a threshold that skips nothing, against ONE oracle attention over the concatenated
sequence; the v2v lists bit-exact against the oracle over three steps - the v2v skip state must survive the interleaved dense calls
(they neither read nor flip the ping-pong lists); the dense calls with few items are split over the keys (host-side split-KV)."""
import pytest
import torch

from helpers import structured_qkv

pytestmark = pytest.mark.gpu


def _recipe(L, att, q, k, v, text_len, scale=None):
    """The README's code, literally; the merge in its last line."""
    query_text, query_video = q[:, :text_len], q[:, text_len:]
    key_text, key_video = k[:, :text_len], k[:, text_len:]
    value_text, value_video = v[:, :text_len], v[:, text_len:]
    att.enable_skip_optimization(enable=False)
    output_t2t, lse_t2t = att(query_text, key_text, value_text, scale, return_softmax_lse=True)
    output_t2v, lse_t2v = att(query_text, key_video, value_video, scale, return_softmax_lse=True)
    output_v2t, lse_v2t = att(query_video, key_text, value_text, scale, return_softmax_lse=True)
    att.enable_skip_optimization(enable=True)
    output_v2v, lse_v2v = att(query_video, key_video, value_video, scale, return_softmax_lse=True)
    out_text, lse_text = L.flash_attn_combine([output_t2t, output_t2v], [lse_t2t, lse_t2v])
    out_video, lse_video = L.flash_attn_combine([output_v2t, output_v2v], [lse_v2t, lse_v2v])
    return torch.cat([out_text, out_video], dim=1), torch.cat([lse_text, lse_video], dim=2)


@pytest.mark.parametrize("vote", ["tile", "half"])
@pytest.mark.parametrize("text_len,video_len", [(77, 1459), (512, 2048)])
def test_text_video_recipe_over_three_steps(text_len, video_len, vote, monkeypatch):
    if vote == "half":
        monkeypatch.setenv("LA_VOTE", "half")
    import liteattention_amd as L
    from oracle import oracle as orc
    B, H, D, thr = 1, 2, 128, -3.0
    S = text_len + video_len
    bm, bn = L.get_tile_sizes(D, 2)
    Qt, Kt = -(-video_len // bm), -(-video_len // bn)
    att = L.LiteAttention(threshold=thr, max_batch_size=B)
    lists_cpu = orc.init_skip_list_ref(B, Qt, Kt, H)
    md_row = orc.expand_must_do_ref([0, 0], bn, Kt + 1)
    for step in range(3):
        q, k, v = structured_qkv(B, S, H, D, seed=40 + step)
        out, lse = _recipe(L, att, q.cuda(), k.cuda(), v.cuda(), text_len)
        assert att._skip_list.shape == (2, B, H, Qt, Kt + 1) and att._phase == (step + 1) % 2      # the dense calls did not touch the v2v state
        qt, qv = q[:, :text_len], q[:, text_len:]
        kt_, kv = k[:, :text_len], k[:, text_len:]
        vt, vv = v[:, :text_len], v[:, text_len:]
        rd, wr = lists_cpu[step % 2], lists_cpu[(step + 1) % 2]
        margins = torch.empty(B, H, Qt, Kt)
        parts = [orc.qkskip_fwd(qt, kt_, vt, block_m=bm, block_n=bn), orc.qkskip_fwd(qt, kv, vv, block_m=bm, block_n=bn),
                 orc.qkskip_fwd(qv, kt_, vt, block_m=bm, block_n=bn),
                 orc.qkskip_fwd(qv, kv, vv, block_m=bm, block_n=bn, read_list=rd, write_list=wr, must_do_list=md_row, thr=thr, margins=margins)]
        # the v2v lists: bit-exact (1e-3 margin rule), and the kernel's list is what the next step reads
        got = att._skip_list[(step + 1) % 2].cpu()
        for h in range(H):
            for m in range(Qt):
                n = int(wr[0, h, m, 0])
                if not (int(got[0, h, m, 0]) == n and torch.equal(got[0, h, m, : n + 1], wr[0, h, m, : n + 1])):
                    mg = margins[0, h, m]
                    assert ((mg[~torch.isnan(mg)] - thr).abs() < 1e-3).any(), (step, h, m)
                    lists_cpu[(step + 1) % 2][0, h, m] = got[0, h, m]      # a borderline tile: follow the kernel from here on
        # merged result = the oracle's parts merged by the merge oracle
        ref_text, lse_ref_text = orc.attention_combine_ref(torch.stack([parts[0][0], parts[1][0]]), torch.stack([parts[0][1], parts[1][1]]).transpose(-1, -2))
        ref_video, lse_ref_video = orc.attention_combine_ref(torch.stack([parts[2][0], parts[3][0]]), torch.stack([parts[2][1], parts[3][1]]).transpose(-1, -2))
        ref = torch.cat([ref_text, ref_video], dim=1)
        lse_ref = torch.cat([lse_ref_text.transpose(1, 2), lse_ref_video.transpose(1, 2)], dim=2)
        tol = 2.0 ** -7 * ref.abs().max().item() + 1e-3          # two bf16 roundings (partials, merge) where one launch has one
        assert (out.float().cpu() - ref).abs().max().item() <= tol, step
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3, step
    assert att.get_skip_fraction() > 0.02                       # the recipe really skipped v2v tiles on this data


def test_recipe_at_a_threshold_that_skips_nothing_is_one_full_attention():
    import liteattention_amd as L
    from oracle import oracle as orc
    B, H, D, text_len, video_len = 2, 3, 128, 200, 1000
    q, k, v = structured_qkv(B, text_len + video_len, H, D, seed=7)
    att = L.LiteAttention(threshold=-60.0, max_batch_size=B)
    bm, bn = L.get_tile_sizes(D, 2)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn)
    for _ in range(2):
        out, lse = _recipe(L, att, q.cuda(), k.cuda(), v.cuda(), text_len)
        assert (out.float().cpu() - o_ref).abs().max().item() <= 2.0 ** -7 * o_ref.abs().max().item() + 1e-3
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    assert att.get_skip_fraction() == 0.0


@pytest.mark.parametrize("D", [128, 256])
def test_recipe_in_e4m3_is_one_full_attention(D):
    """The same recipe on e4m3 inputs (bf16 partials, fp32 LSE; the dense calls with few items split over the keys as for bf16): at a threshold
    that skips nothing, one oracle attention over the concatenated sequence in the kernel's form of P, inside the fp8 bound; LSE to 1e-3 (the
    default form: fp32 row sums of the un-rounded P)."""
    import liteattention_amd as L
    from oracle import oracle as orc
    F8 = torch.float8_e4m3fn
    B, H, text_len, video_len = 1, 3, 200, 1500
    q, k, v = [x.to(F8) for x in structured_qkv(B, text_len + video_len, H, D, seed=9, dtype=torch.float32)]
    att = L.LiteAttention(threshold=-60.0, max_batch_size=B)
    bm, bn = L.get_tile_sizes(D, 1)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round="fp8")
    for _ in range(2):
        out, lse = _recipe(L, att, q.cuda(), k.cuda(), v.cuda(), text_len)
        assert out.dtype == torch.bfloat16
        assert (out.float().cpu() - o_ref).abs().max().item() <= 0.05 * o_ref.abs().max().item() + 2e-2
        assert (lse.cpu() - lse_ref).abs().max().item() <= 1e-3
    assert att.get_skip_fraction() == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,n", [(1, 512, 9000, 8, 8, -1), (2, 300, 5000, 4, 2, -1), (1, 100, 3000, 2, 2, 5), (3, 64, 700, 2, 1, 3)])
def test_split_kv_of_dense_launches_with_few_items(B, Sq, Sk, H, Hk, n, dtype):
    """Host-side split-KV (reference: get_num_splits, flash_api.cpp:437-465; heuristics.h:25-58): ``num_splits=-1`` decides by the
    reference's rule, ``num_splits=n`` forces n (0 and 1: no split, as in the reference's default build); the result is the unsplit
    launch's up to the merge's rounding, and matches the oracle."""
    import liteattention_amd as L
    from liteattention_amd.flash_attn_interface import _num_splits
    from oracle import oracle as orc
    g = torch.Generator().manual_seed(Sq + Sk)
    q = torch.randn(B, Sq, H, 128, generator=g).to(dtype)
    k = torch.randn(B, Sk, Hk, 128, generator=g).to(dtype)
    v = torch.randn(B, Sk, Hk, 128, generator=g).to(dtype)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    chosen = _num_splits(B, H, Sq, Sk, 128, 2, max(n, 0))
    assert chosen > 1                                           # these shapes leave most of the device idle unsplit
    o1, l1 = L.flash_attn_func(qd, kd, vd, return_softmax_lse=True)                     # num_splits = 1: the reference's default
    o2, l2 = L.flash_attn_func(qd, kd, vd, num_splits=n, return_softmax_lse=True)
    bm, bn = L.get_tile_sizes(128, 2)
    o_ref, lse_ref, _ = orc.qkskip_fwd(q, k, v, block_m=bm, block_n=bn, p_round="f16" if dtype == torch.float16 else True)
    for o, l, extra in ((o1, l1, 0.0), (o2, l2, 2.0 ** -9)):
        assert o.shape == q.shape and o.dtype == dtype and l.shape == (B, H, Sq) and l.is_contiguous()
        assert (o.float().cpu() - o_ref).abs().max().item() <= (2.0 ** -8 + extra) * o_ref.abs().max().item() + 1e-3
        assert (l.cpu() - lse_ref).abs().max().item() <= 1e-3
    o3 = L.flash_attn_func(qd, kd, vd, num_splits=n if n > 0 else chosen)
    assert torch.equal(o3, o2)
    assert torch.equal(L.flash_attn_func(qd, kd, vd, num_splits=0), o1)
    # K / V with a padded batch stride cannot be seen as packed rows: the unsplit launch runs instead, same result as num_splits = 1
    if B > 1:
        kp = torch.empty(B, Sk + 8, Hk, 128, dtype=dtype, device="cuda")[:, :Sk]
        kp.copy_(kd)
        o4 = L.flash_attn_func(qd, kp, vd, num_splits=-1)
        assert torch.equal(o4, o1)
    with pytest.raises(NotImplementedError):
        rd = orc.init_skip_list_ref(B, -(-Sq // bm), -(-Sk // bn), H)
        L.flash_attn_func(qd, kd, vd, num_splits=2, attn_read_list=rd[0].cuda(), attn_write_list=rd[1].cuda())


def test_combine_of_separate_partials_equals_the_stacked_form():
    import liteattention_amd as L
    torch.manual_seed(3)
    B, S, H, D, n = 2, 333, 3, 128, 4
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        outs = [torch.randn(B, S, H, D, device="cuda").to(dt) for _ in range(n)]
        lses = [torch.randn(B, H, S, device="cuda") for _ in range(n)]
        lses[1][0, 1, 5] = float("-inf")
        a, la = L.flash_attn_combine(outs, lses)
        b, lb = L.flash_attn_combine(torch.stack(outs), torch.stack(lses))
        assert torch.equal(a, b) and torch.equal(la, lb) and a.dtype == dt
    with pytest.raises(RuntimeError):
        L.flash_attn_combine(outs * 3, lses * 3)                # more than 8 partials: stack them
