"""GPU: tensors beyond 4 GiB. B = 6 at the headline shape puts the last batch entry of q / k / v / o 3.9-4.6 GB behind the base pointer:
every offset on the path must be 64-bit (C-ABI strides are int64; the hand-scheduled bodies build row addresses with v_mad_u64_u32,
the tile-address table holds 64-bit tile addresses, the list offset is (bh * q_tiles + m) * (k_tiles + 1) in 64 bits). bf16 with
imposed 42 % lists through LiteAttention (lists grown to batch 6: 673 MB), fp8 dense; sampled rows of batch entries 0, 3 and 5 against
fp32 torch, whole output finite, list fixed point."""
import pytest
import torch

pytestmark = pytest.mark.gpu
S, H, D, B = 75600, 40, 128, 6


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_batch_of_six_at_the_headline_shape(dtype):
    import liteattention_amd as L
    from tools import selfcheck as sc
    fp8 = dtype == "fp8"
    g = torch.Generator(device="cuda").manual_seed(7)
    q, k, v = [torch.randn(B, S, H, D, device="cuda", generator=g, dtype=torch.bfloat16) for _ in range(3)]
    assert q.numel() * q.element_size() > 2 ** 32
    if fp8:
        q, k, v = [x.to(torch.float8_e4m3fn) for x in (q, k, v)]
        out, lse = L.flash_attn_func(q, k, v, return_softmax_lse=True)
        read, (bm, bn) = None, L.get_tile_sizes(D, 1)
        tol = dict(o_rtol=0.05, o_atol=1e-3, lse_atol=2.5e-3)
    else:
        bm, bn = L.get_tile_sizes(D, 2)
        qt, kt = -(-S // bm), -(-S // bn)
        att = L.LiteAttention(max_batch_size=B)
        att.threshold = float("-inf")
        att._get_read_write_lists(q, k)
        att._phase = 0
        sc.impose_lists(att, sc.banded_rows(qt, kt, bm, bn, 0.42))
        read = att._skip_list[0].clone()
        att._skip_list[1].fill_(-7)
        out, lse = att(q, k, v, return_softmax_lse=True)
        n = int(read[..., 0].max().item())
        live = torch.arange(n + 1, device="cuda") <= read[..., 0:1]
        assert bool(((att._skip_list[1][..., : n + 1] == read[..., : n + 1]) | ~live).all())      # fixed point at thr = -inf, every batch entry
        tol = dict(o_rtol=2.0 ** -8, o_atol=1e-4, lse_atol=2e-4)
    assert bool(torch.isfinite(out.float()).all()) and bool(torch.isfinite(lse).all())
    for b in (0, 3, 5):
        res = sc.sampled_row_check(q, k, v, out, lse, read, bm, bn, heads=(0, 39), n_rows=128, batch=b, **tol)
        assert res["ok"], (b, res)
