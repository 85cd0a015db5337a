"""GPU: the reference's OWN forward grid, as its test file lays it out (hopper/tests/test_flash_attn.py:52-125, `test_lite_attn_output`:
every (seqlen_q, seqlen_k) of its list x every compiled head dim x bf16 / fp16 / e4m3, batch 9 (2 above 2048 keys), 6 heads, through
`LiteAttention(enable_skipping=True, threshold=-3.0)` exactly as that test constructs it; the test's body stops after building its inputs
in the reference, so the assertions here are the ones its sibling `test_flash_attn_output` makes, :266-296):

  call 1 (no lists yet: every tile is listed)  - the reference's rule against eager torch evaluated on the device like the reference does:
         |out - out_ref| <= 2 |out_pt - out_ref| + fwd_atol, and the LSE against torch.logsumexp;
  call 2 (the lists call 1 wrote, same inputs) - output, LSE and the written lists against the C oracle walking the same read lists
         (lists bit for bit unless a tile's vote margin is below 1e-3: tests/test_gpu_parity.py::_compare_lists).

V_colmajor of the reference's grid: `mha_fwd` refuses a V whose last dimension is not contiguous (flash_api.cpp:728), here as there
(tests/test_gpu_parity.py, error paths); causal / local / softcap / qv / pack_gqa / num_splits are compiled out of the reference's
LiteAttention build (hopper/setup.py:47-63) and raise typed errors here."""
import pytest
import torch

from helpers import fp8_lse_tol, fp8_p_round
from test_gpu_parity import _compare_lists

pytestmark = pytest.mark.gpu

SEQLENS = [(1, 1), (64, 128), (128, 192), (256, 256), (239, 1), (799, 3), (113, 203), (113, 128), (128, 217), (113, 211), (108, 256),
           (256, 512), (384, 256), (640, 128), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023), (4096, 4096), (4224, 4224)]      # :61-85
HDIMS = [64, 96, 128, 192, 256]                                                                                                       # COMPILED_HDIMS :44-51
F8 = torch.float8_e4m3fn
THR = -3.0


def _inputs(seqlen_q, seqlen_k, d, dtype):
    """test_flash_attn.py:91-115: seed 0, batch 9 (2 for long keys), 6 heads, randn rounded to the kernel's type."""
    torch.random.manual_seed(0)
    B, H = (9 if seqlen_k <= 2048 else 2), 6
    dtype_ref = torch.bfloat16 if dtype == F8 else dtype
    q, k, v = [torch.randn(B, s, H, d, device="cuda", dtype=dtype_ref).to(dtype) for s in (seqlen_q, seqlen_k, seqlen_k)]
    return B, H, q, k, v, dtype_ref


def _second_call_against_the_oracle(L, orc, att, q, k, v, B, H, d, dtype, tol_out, tol_lse, p_round):
    BM, BN = L.get_tile_sizes(d, q.element_size())
    Qt, Kt = -(-q.shape[1] // BM), -(-k.shape[1] // BN)
    rd_idx = att._phase
    out, lse = att(q, k, v, return_softmax_lse=True)
    rd, wr = att._skip_list[rd_idx].cpu(), att._skip_list[1 - rd_idx].cpu()
    wr_orc = torch.zeros_like(wr)
    margins = torch.empty(B, H, Qt, Kt)
    md_row = orc.expand_must_do_ref([0, 0], BN, Kt + 1) if Kt >= 2 else None      # a one-tile row has no room for [2, 0, 0]: no must-do list
    o_ref, lse_ref, n_tiles = orc.qkskip_fwd(q.cpu(), k.cpu(), v.cpu(), block_m=BM, block_n=BN, read_list=rd, write_list=wr_orc,
                                             must_do_list=md_row, thr=THR, margins=margins, p_round=p_round)
    assert n_tiles == orc.listed_tiles(rd[:B])
    assert (out.float().cpu() - o_ref).abs().max().item() <= tol_out(o_ref)
    assert (lse.cpu() - lse_ref).abs().max().item() <= tol_lse
    bad, border = _compare_lists(orc, rd, wr, wr_orc, margins, THR, B)
    assert bad == 0, f"{bad} rows differ from the oracle with no borderline tile"
    assert border <= max(2, B * H * Qt // 100)          # rows with a vote within 1e-3 of the threshold may go either way


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", HDIMS)
@pytest.mark.parametrize("seqlen_q,seqlen_k", SEQLENS)
def test_lite_attn_output(seqlen_q, seqlen_k, d, dtype):
    import liteattention_amd as L
    from oracle import oracle as orc
    B, H, q, k, v, _ = _inputs(seqlen_q, seqlen_k, d, dtype)
    # test_flash_attn.py:103 constructs it with the default max_batch_size = 4, which its own batch of 9 would trip over at the first call
    # (lite_attention.py:158): the bound is given here
    att = L.LiteAttention(enable_skipping=True, threshold=THR, max_batch_size=B)
    out, lse = att(q, k, v, return_softmax_lse=True)
    assert out.dtype == dtype and out.shape == q.shape and lse.shape == (B, H, seqlen_q)
    out_ref, lse_ref = orc.attention_dense_ref(q.float(), k.float(), v.float())      # eager torch on the device, fp32 (:224-238)
    out_pt, _ = orc.attention_dense_ref(q, k, v, upcast=False, reorder_ops=True)     # the same in the kernel's type (:239-254)
    err, bound = (out.float() - out_ref).abs().max().item(), orc.dense_tolerance(out_ref, out_pt)
    assert err <= bound, (err, bound)                                                # :296
    assert (lse - lse_ref).abs().max().item() <= 1e-3
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    _second_call_against_the_oracle(L, orc, att, q, k, v, B, H, d, dtype, lambda o: ulp * o.abs().max().item() + 1e-3, 1e-3,
                                    "f16" if dtype == torch.float16 else True)


@pytest.mark.parametrize("form", ["encoded", "exact"])
@pytest.mark.parametrize("seqlen_q,seqlen_k", SEQLENS)
def test_lite_attn_output_e4m3(seqlen_q, seqlen_k, form, monkeypatch):
    """The e4m3 column of the same grid at head_dim 128 (the fp8 kernels' head dim; fp8 is compiled out of the reference's default build,
    hopper/setup.py:56): the block-scaled encoding of P (LA_FLAG_FP8_ENCODED_P) and the default form (the reference's arithmetic), against the C oracle
    in the same form, with the fp8 tolerances of tests/test_gpu_fp8.py."""
    import liteattention_amd as L
    from oracle import oracle as orc
    monkeypatch.delenv("LA_FP8_P", raising=False)
    if form == "encoded":
        monkeypatch.setenv("LA_FP8_P", "encoded")
    B, H, q, k, v, _ = _inputs(seqlen_q, seqlen_k, 128, F8)
    att = L.LiteAttention(enable_skipping=True, threshold=THR, max_batch_size=B)
    out, lse = att(q, k, v, return_softmax_lse=True)
    assert out.dtype == torch.bfloat16 and out.shape == q.shape                     # bf16 out for fp8 inputs, flash_api.cpp:859
    BM, BN = L.get_tile_sizes(128, 1)
    o8, lse8, _ = orc.qkskip_fwd(q.cpu(), k.cpu(), v.cpu(), block_m=BM, block_n=BN, p_round=fp8_p_round())
    tol = lambda o: 0.05 * o.abs().max().item() + 2e-2
    assert (out.float().cpu() - o8).abs().max().item() <= tol(o8)
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()
    _second_call_against_the_oracle(L, orc, att, q, k, v, B, H, 128, F8, tol, fp8_lse_tol(), fp8_p_round())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mha_type", ["mha", "mqa", "gqa"])
@pytest.mark.parametrize("d", HDIMS)
@pytest.mark.parametrize("seqlen_q,seqlen_k", SEQLENS)
def test_flash_attn_output(seqlen_q, seqlen_k, d, mha_type, dtype):
    """The non-causal, dv == d subset of the reference's dense test (test_flash_attn.py:126-296; mha / mqa / gqa with 6 query heads over
    6 / 1 / 2 K/V heads) through `flash_attn_func`, held to the rule of its line 296 with both eager evaluations made on the device."""
    import liteattention_amd as L
    from oracle import oracle as orc
    torch.random.manual_seed(0)
    B, H = (9 if seqlen_k <= 2048 else 2), 6
    Hk = H if mha_type == "mha" else (2 if mha_type == "gqa" else 1)
    q = torch.randn(B, seqlen_q, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, seqlen_k, Hk, d, device="cuda", dtype=dtype)
    v = torch.randn(B, seqlen_k, Hk, d, device="cuda", dtype=dtype)
    out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    out_ref, lse_ref = orc.attention_dense_ref(q.float(), k.float(), v.float())
    out_pt, _ = orc.attention_dense_ref(q, k, v, upcast=False, reorder_ops=True)
    err, bound = (out.float() - out_ref).abs().max().item(), orc.dense_tolerance(out_ref, out_pt)
    assert err <= bound, (err, bound)
    assert (lse - lse_ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("mha_type", ["mha", "mqa", "gqa"])
@pytest.mark.parametrize("seqlen_q,seqlen_k", SEQLENS)
def test_flash_attn_output_e4m3(seqlen_q, seqlen_k, mha_type, monkeypatch):
    """The e4m3 column of that test (head_dim 128; descales drawn as at :214: rand(batch, K/V heads) * 2), in the reference's arithmetic
    (the default form) against the C oracle with the same descales."""
    import liteattention_amd as L
    from oracle import oracle as orc
    monkeypatch.delenv("LA_FP8_P", raising=False)              # the default form IS the reference's arithmetic
    torch.random.manual_seed(0)
    B, H = (9 if seqlen_k <= 2048 else 2), 6
    Hk = H if mha_type == "mha" else (2 if mha_type == "gqa" else 1)
    q = torch.randn(B, seqlen_q, H, 128, device="cuda", dtype=torch.bfloat16).to(F8)
    k = torch.randn(B, seqlen_k, Hk, 128, device="cuda", dtype=torch.bfloat16).to(F8)
    v = torch.randn(B, seqlen_k, Hk, 128, device="cuda", dtype=torch.bfloat16).to(F8)
    qd, kd, vd = [torch.rand(B, Hk, device="cuda", dtype=torch.float32) * 2 for _ in range(3)]
    out, lse = L.flash_attn_func(q, k, v, q_descale=qd, k_descale=kd, v_descale=vd, return_softmax_lse=True)
    assert out.dtype == torch.bfloat16
    BM, BN = L.get_tile_sizes(128, 1)
    o8, lse8, _ = orc.qkskip_fwd(q.cpu(), k.cpu(), v.cpu(), block_m=BM, block_n=BN, p_round=fp8_p_round(),
                                 q_descale=qd.cpu(), k_descale=kd.cpu(), v_descale=vd.cpu())
    assert (out.float().cpu() - o8).abs().max().item() <= 0.05 * o8.abs().max().item() + 2e-2
    assert (lse.cpu() - lse8).abs().max().item() <= fp8_lse_tol()


VARLEN_SEQLENS = [(1, 1), (1, 3), (2, 1), (511, 1), (3, 513), (64, 128), (128, 128), (256, 256), (113, 203), (128, 217), (113, 211),
                  (108, 256), (256, 512), (307, 256), (640, 128), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023), (2048, 2048),
                  (4096, 4096)]                                                                                                  # :388-413


def _random_lengths(max_seqlen, batch, zero_lengths):
    """test_util.py:9-29, mode "random": lengths in [max(0 or 1, max - 20), max]; with zero_lengths every fifth sequence and the last are empty."""
    lengths = torch.randint(max(0 if zero_lengths else 1, max_seqlen - 20), max_seqlen + 1, (batch,))
    if zero_lengths:
        lengths[::5] = 0
        lengths[-1] = 0
    return lengths.tolist()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mha_type", ["mha", "mqa", "gqa"])
@pytest.mark.parametrize("d", HDIMS)
@pytest.mark.parametrize("seqlen_q,seqlen_k", VARLEN_SEQLENS)
def test_flash_attn_varlen_output(seqlen_q, seqlen_k, d, mha_type, dtype):
    """The non-causal, dv == d, add_unused_qkv = False subset of the reference's varlen test (test_flash_attn.py:363-560): random query
    lengths, random key lengths with EMPTY key sequences (every fifth and the last, test_util.py:20-25), packed tensors + cu_seqlens
    through `flash_attn_varlen_func` (one launch). Rule of :296 over the whole batch; a sequence without keys gives zeros
    (flash_api.cpp:1241-1245)."""
    import liteattention_amd as L
    from oracle import oracle as orc
    torch.random.manual_seed(seqlen_q + seqlen_k + d)
    B, H = (9 if seqlen_q <= 2048 else 2), 6
    Hk = H if mha_type == "mha" else (2 if mha_type == "gqa" else 1)
    q = torch.randn(B, seqlen_q, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, seqlen_k, Hk, d, device="cuda", dtype=dtype)
    v = torch.randn(B, seqlen_k, Hk, d, device="cuda", dtype=dtype)
    lq, lk = _random_lengths(seqlen_q, B, False), _random_lengths(seqlen_k, B, True)
    cu = lambda ls: torch.tensor([0] + list(torch.tensor(ls).cumsum(0)), dtype=torch.int32, device="cuda")
    q_un = torch.cat([q[b, :lq[b]] for b in range(B)])
    k_un = torch.cat([k[b, :lk[b]] for b in range(B)])
    v_un = torch.cat([v[b, :lk[b]] for b in range(B)])
    out_un = L.flash_attn_varlen_func(q_un, k_un, v_un, cu(lq), cu(lk), max(lq), max(lk))
    assert out_un.shape == q_un.shape and out_un.dtype == dtype
    err = pt_err = 0.0
    refs, off = [], 0
    for b in range(B):
        o = out_un[off:off + lq[b]].float()
        off += lq[b]
        if lk[b] == 0:
            assert (o == 0).all(), f"sequence {b} has no keys: its rows must be zero"
            continue
        o_ref, _ = orc.attention_dense_ref(q[b:b + 1, :lq[b]].float(), k[b:b + 1, :lk[b]].float(), v[b:b + 1, :lk[b]].float())
        o_pt, _ = orc.attention_dense_ref(q[b:b + 1, :lq[b]], k[b:b + 1, :lk[b]], v[b:b + 1, :lk[b]], upcast=False, reorder_ops=True)
        err = max(err, (o - o_ref[0]).abs().max().item())
        pt_err = max(pt_err, (o_pt.float() - o_ref).abs().max().item())
        refs.append(o_ref)
    fwd_atol = max(2 * (r + 0.3 - 0.3 - r).abs().max().item() for r in refs) if refs else 0.0
    assert err <= 2 * pt_err + fwd_atol, (err, pt_err, fwd_atol)


@pytest.mark.parametrize("d", [32, 40, 64, 80, 96, 128, 160, 192, 224, 256])
@pytest.mark.parametrize("seqlen_q,seqlen_k", [(1, 239), (239, 1), (3, 799), (799, 3), (1024, 128), (97, 97), (128, 128), (200, 200),
                                               (256, 256), (257, 257), (384, 384), (512, 512), (768, 768), (1024, 1024), (2048, 2048)])
def test_flash_attn_race_condition(seqlen_q, seqlen_k, d):
    """The forward half of test_flash_attn.py:1116-1166 (non-causal; its head dims that are multiples of 8 - 59 and 111 fail the reference's own
    `head_size % 8` check, flash_api.cpp:854): batch 60, 4 heads, under 70 GiB of allocated memory, the same call again and again must give
    the same bits (100 repetitions here; the skip-list forms of the same screen: tests/test_gpu_round2.py, tests/test_gpu_head_dims.py)."""
    import liteattention_amd as L
    torch.random.manual_seed(0)
    try:
        dummy = torch.empty(70 * 1024 ** 3, dtype=torch.uint8, device="cuda")      # "simulate under memory load" (:1147)
    except RuntimeError:                                                           # a box with less free memory: the screen still runs
        dummy = None
    q = torch.randn(60, seqlen_q, 4, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(60, seqlen_k, 4, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(60, seqlen_k, 4, d, device="cuda", dtype=torch.bfloat16)
    out0, lse0 = L.flash_attn_func(q, k, v, return_softmax_lse=True)
    for _ in range(100):
        out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
        assert torch.equal(out, out0) and torch.equal(lse, lse0)
    del dummy
