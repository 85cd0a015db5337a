"""CPU: host-side logic of the adapters (no device work)."""
import pytest
import torch

import liteattention_amd as L
from liteattention_amd.compat import blockmask_to_rows
from oracle import oracle as orc


def test_blockmask_rows_walk_exactly_the_kept_tiles():
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        kt = int(torch.randint(1, 40, (1,), generator=g))
        mask = torch.rand(7, kt, generator=g) < 0.5
        mask[:, int(torch.randint(0, kt, (1,), generator=g))] = True      # at least one kept tile per row
        rows = blockmask_to_rows(mask)
        for m, row in enumerate(rows):
            kept = [j for j in range(kt - 1, -1, -1) if mask[m, j]]
            assert orc.walk_tiles(row + [0, 0]) == kept
            assert row[0] % 2 == 0 and row[0] <= kt + (kt % 2)
    with pytest.raises(ValueError):
        blockmask_to_rows(torch.tensor([[1, 0], [0, 0]]))


def test_blockmask_skip_lists_shape_and_broadcast():
    mask = torch.tensor([[1, 1, 0, 1], [0, 0, 1, 0], [1, 1, 1, 1]])
    lists = L.blockmask_to_skip_lists(mask, batch=2, heads=3, device="cpu")
    assert lists.shape == (2, 2, 3, 3, 5) and lists.dtype == torch.int32
    assert torch.equal(lists[0], lists[1])
    assert lists[0, 1, 2, 0].tolist() == [4, 3, 3, 1, 0] and lists[0, 0, 0, 1].tolist() == [2, 2, 2, 0, 0]
    per_head = mask[None, None].repeat(2, 3, 1, 1).clone()
    per_head[1, 2, 1] = torch.tensor([1, 0, 0, 0])
    lists2 = L.blockmask_to_skip_lists(per_head, batch=2, heads=3, device="cpu")
    assert lists2[0, 1, 2, 1].tolist() == [2, 0, 0, 0, 0] and torch.equal(lists2[0, 0], lists[0, 0])


def test_adapters_reject_options_outside_the_path():
    q = torch.zeros(1, 64, 1, 128, dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        L.fa2_flash_attn_func(q, q, q, causal=True)
    with pytest.raises(NotImplementedError):
        L.fa2_flash_attn_func(q, q, q, dropout_p=0.1)
    with pytest.raises(NotImplementedError):
        L.flash_attn_varlen_func(q[0], q[0], q[0], [0, 64], [0, 64], window_size=(8, 0))
    with pytest.raises(RuntimeError):
        L.flash_attn_varlen_func(q[0], q[0], q[0], [0, 64], [0, 32, 64])
    with pytest.raises(ValueError):
        L.flash_blocksparse_attn_func(q, q, q, torch.ones(2, 2))
