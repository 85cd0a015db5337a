"""CPU: the workload generators of bench.py (imposed-sparsity lists, executed-FLOP accounting, q-window planning) —
the numbers the bench line is built from."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def test_banded_rows_hit_the_requested_sparsity_and_are_valid_lists():
    S, bm, bn = 75600, 256, 64
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    for s in bench.SPARSITIES:
        rows = bench.banded_rows(Qt, Kt, bm, bn, s)
        listed = bench.listed_tiles_of_rows(rows)
        assert abs((1 - listed / (Qt * Kt)) - s) < 0.005
        for r in rows[:: max(1, Qt // 17)].tolist() + [rows[-1].tolist()]:
            tiles = orc.walk_tiles(r)
            assert tiles[0] == Kt - 1                               # the first walked tile carries the seqlen mask
            assert tiles == sorted(set(tiles), reverse=True)        # descending, no duplicates
            assert 0 <= min(tiles) and len(tiles) == (r[1] - r[2] + 1) + ((r[3] - r[4] + 1) if r[0] == 4 else 0)


def test_executed_flops_counts_ragged_edges_at_their_real_size():
    S, bm, bn, D, H = 1000, 256, 64, 128, 3
    Qt, Kt = math.ceil(S / bm), math.ceil(S / bn)
    rows = bench.banded_rows(Qt, Kt, bm, bn, 0.0)
    assert bench.executed_flops(rows, H, 1, S, S, bm, bn, D) == 4.0 * H * S * S * D          # dense: exactly 4*H*S^2*D
    half = bench.banded_rows(Qt, Kt, bm, bn, 0.5)
    fl = bench.executed_flops(half, H, 1, S, S, bm, bn, D)
    assert 0.4 < fl / (4.0 * H * S * S * D) < 0.62


def test_impose_lists_writes_both_ping_pong_buffers():
    class A:                                                   # stand-in with the attribute bench touches
        _skip_list = torch.full((2, 1, 2, 4, 9), 7, dtype=torch.int32)
    rows = bench.banded_rows(4, 8, 256, 64, 0.5)
    bench.impose_lists(A, rows)
    assert torch.equal(A._skip_list[0], A._skip_list[1])
    assert torch.equal(A._skip_list[0, 0, 1, :, :5], rows) and int(A._skip_list[..., 5:].abs().sum()) == 0
