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
    assert rows[-1][1] == 0.0 and rows[-1][2] < 1e-6
