"""GPU: module-level drop-in (SURVEY §8b "callers"): a Wan2.x-style self-attention block built on
`from lite_attention import LiteAttention` per the reference's README recipe (README.md:268-323)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_wan_style_block_runs_and_skipping_stays_close_to_dense():
    import wan_self_attention_demo as demo
    rows = demo.run(frames=4, height=16, width=20, heads=3, steps=5, threshold=-6.0, verbose=False)   # S = 1280
    skipped = [r[1] for r in rows]
    assert skipped == sorted(skipped)                      # sparsity only grows between resets (Appendix A.3)
    assert skipped[-1] > 0.05                              # structured video-like input: something is skipped
    assert all(r[2] < 2e-2 for r in rows)                  # thr = -6: contributions below 2^-6 of the running max dropped
    # threshold very negative: nothing may be skipped and the block equals the dense block to bf16 round-off
    rows = demo.run(frames=4, height=16, width=20, heads=3, steps=2, threshold=-60.0, verbose=False)
    # (2^-7: the dense block's attention has few items here - 15 on 256 compute units - and is split over the keys on the host, so its
    # result went through one more bf16 rounding, the merge's, than the list-walking launch it is compared with)
    assert rows[-1][1] == 0.0 and rows[-1][2] < 2.0 ** -7


def test_wan_cross_attention_through_the_flash_attn_import_names():
    """The text cross-attention of the block: ragged prompt lengths, packed tensors, `import flash_attn_interface` from compat_shims/."""
    import torch
    import wan_self_attention_demo as demo
    shims = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat_shims")
    sys.path.insert(0, shims)
    try:
        torch.manual_seed(0)
        blk = demo.WanLikeCrossAttention(3 * 128, 3).cuda()
        x = torch.randn(2, 700, 3 * 128, device="cuda")
        ctx = torch.randn(2, 512, 3 * 128, device="cuda")
        lens = [77, 512]
        with torch.no_grad():
            y, y_ref = blk(x, ctx, lens), blk.reference(x, ctx, lens)
        assert (y - y_ref).abs().max().item() <= 2e-2 * y_ref.abs().max().item()
    finally:
        sys.path.remove(shims)
        for n in [m for m in sys.modules if m == "flash_attn" or m.startswith("flash_attn.") or m == "flash_attn_interface"]:
            del sys.modules[n]


def test_two_steps_capture_into_a_hip_graph_and_replay():
    """HIP graphs instead of a tracing compiler: `LiteAttention.__call__` makes no host sync and allocates only through
    torch's allocator, so a PAIR of denoising steps (one per ping-pong phase) captures into one graph; every replay
    advances the skip state by two steps exactly as two eager calls do."""
    import liteattention_amd as L
    from helpers import structured_qkv
    q, k, v = [x.cuda() for x in structured_qkv(1, 1536, 2, 128, seed=11)]
    eager = L.LiteAttention(threshold=-3.0, max_batch_size=1)
    graphed = L.LiteAttention(threshold=-3.0, max_batch_size=1)
    # side stream warm-up (allocates the lists), then rewind the state so both objects start equal
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        graphed(q, k, v)
    torch.cuda.current_stream().wait_stream(s)
    graphed._skip_list.copy_(L.LiteAttention.init_skip_list(1, 1536, 2, 128, False, torch.bfloat16, "cuda"))
    graphed._phase = 0
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o_a = graphed(q, k, v)
        o_b = graphed(q, k, v)
    graphed._skip_list.copy_(L.LiteAttention.init_skip_list(1, 1536, 2, 128, False, torch.bfloat16, "cuda"))
    for rep in range(3):
        g.replay()
        e_a = eager(q, k, v)
        e_b = eager(q, k, v)
        torch.cuda.synchronize()
        assert torch.equal(o_a, e_a) and torch.equal(o_b, e_b), rep
        assert torch.equal(graphed._skip_list, eager._skip_list), rep
    assert eager.get_skip_fraction() > 0.02
