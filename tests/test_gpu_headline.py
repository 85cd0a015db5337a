"""GPU: correctness gate on the HEADLINE shape — BASELINE.json configs[2]/[4]: B=1, S=75 600, H=40, D=128, bf16 and fp8 —
through ``LiteAttention.__call__`` with the lists ``bench.py`` times (imposed 42 % / 77 % bands) and with the initial full list.

This is the shape where the persistent grid, all 11 840 (head, q-tile) items, the eight per-XCD ticket queues and the
stealing path engage; the CPU oracle cannot finish it, so the checks are size-independent properties (reference shape grid:
hopper/tests/test_flash_attn.py:152-177; property style of test_gpu_parity.py::test_full_size_properties_c2):

  * `out` pre-filled with NaN stays NaN-free: every item was processed and every row stored;
  * the write list equals the read list at thr = -inf (nothing new may be dropped; bit-exact);
  * dynamic (ticket queues, persistent workgroups) == static (one workgroup per item, XCD map) bit-exactly: O, LSE, lists;
  * >= 256 sampled query rows per checked head against an fp32 torch attention over exactly the listed keys:
        bf16: |O - ref| <= 2^-8 max|ref| + 1e-4      fp8: |O - ref| <= 0.05 max|ref| + 1e-3 (P is an 8-bit quantity)
        LSE:  |LSE - ref| <= 2e-4 (bf16, and fp8 in its default form - the reference's fp32 row sums: one missing 64-key tile of a 43 k-key row moves the LSE
              by 1.5e-3, so a skipped, doubled or mis-masked tile cannot hide); fp8 under LA_FLAG_FP8_ENCODED_P / LA_FLAG_FP8_MFMA_ROWSUM <= 2.5e-3: the row sums are those
              of the ENCODED P, whose noise averages out over a long row but whose bias (about -3e-4 / -7e-4) does not. The walk is
              the same code in all three fp8 forms.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

S, H, D = 75600, 40, 128
F8 = torch.float8_e4m3fn


@pytest.fixture(scope="module")
def qkv():
    g = torch.Generator(device="cuda").manual_seed(1234)
    return [torch.randn(1, S, H, D, device="cuda", generator=g, dtype=torch.float32).to(torch.bfloat16) for _ in range(3)]


@pytest.mark.parametrize("dtype", ["bf16", "fp8", "fp8-exact-exp", "fp8-exact-rowsum"])
@pytest.mark.parametrize("sparsity", [0.42, 0.77, None])
def test_headline_shape(qkv, dtype, sparsity, monkeypatch):
    import liteattention_amd as L
    monkeypatch.delenv("LA_FP8_P", raising=False)
    exact_rowsum = dtype == "fp8-exact-rowsum"                            # the default form: the reference's arithmetic, fp32 row sums
    if exact_rowsum:
        dtype = "fp8"
    elif dtype == "fp8-exact-exp":                                        # LA_FLAG_FP8_MFMA_ROWSUM: v_exp_f32 + hardware e4m3 rounding of P, row sums of the rounded P
        monkeypatch.setenv("LA_FP8_P", "mfma_rowsum")
        dtype = "fp8"
    elif dtype == "fp8":                                                  # LA_FLAG_FP8_ENCODED_P: the block-scaled byte encoding
        monkeypatch.setenv("LA_FP8_P", "encoded")
    from tools import selfcheck as sc
    from liteattention_amd.flash_attn_interface import mha_fwd
    fp8 = dtype == "fp8"
    q, k, v = [x.to(F8) for x in qkv] if fp8 else qkv
    bm, bn = L.get_tile_sizes(D, 1 if fp8 else 2)
    qt, kt = -(-S // bm), -(-S // bn)
    assert H * qt > 40 * 256                                              # many items per CU: the persistent loop re-iterates

    att = L.LiteAttention(max_batch_size=1)
    att.threshold = float("-inf")
    att._get_read_write_lists(q, k)                                       # allocate [2, 1, H, Qt, Kt+1]; initial full list
    att._phase = 0
    if sparsity is not None:
        sc.impose_lists(att, sc.banded_rows(qt, kt, bm, bn, sparsity))
    read = att._skip_list[0].clone()
    att._skip_list[1].fill_(-7)                                           # the kernel must write every row of the write list

    # dynamic work distribution (what LiteAttention.__call__ runs)
    out, lse = att(q, k, v, return_softmax_lse=True)
    assert torch.equal(att._skip_list[0], read)                           # read list untouched
    wr = att._skip_list[1]
    n = int(read[..., 0].max().item())
    live = torch.arange(n + 1, device="cuda") <= read[..., 0:1]           # entries 0..L of every row (beyond L: don't-care)

    def same_rows(a, b):
        return bool(((a[..., : n + 1] == b[..., : n + 1]) | ~live).all())
    assert same_rows(wr, read)                                            # fixed point at thr = -inf, bit-exact
    assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(lse).all())

    # static map, into a NaN-prefilled output: bit-identical
    out_s = torch.full_like(out, float("nan"))
    wr_s = torch.full_like(read, -7)
    must_do = torch.zeros(kt + 1, dtype=torch.int32, device="cuda")
    must_do[0] = 2
    o2, lse_s, *_ = mha_fwd(q, k, v, out=out_s, attn_read_list=read, attn_must_do_list=must_do, attn_write_list=wr_s,
                            thr=float("-inf"), _must_do_is_1d=True, _static_sched=True)
    assert o2.data_ptr() == out_s.data_ptr()
    assert bool(torch.isfinite(out_s.float()).all())                      # no row left at its NaN prefill
    assert torch.equal(out_s, out) and torch.equal(lse_s, lse)
    assert same_rows(wr_s, wr)

    tol = dict(o_rtol=0.05, o_atol=1e-3, lse_atol=2e-4 if exact_rowsum else 2.5e-3) if fp8 else dict(o_rtol=2.0 ** -8, o_atol=1e-4)
    res = sc.sampled_row_check(q, k, v, out, lse, read, bm, bn, heads=(0, 17, 39), n_rows=256, **tol)
    assert res["ok"], res
    assert res["rows"] >= 3 * 256
