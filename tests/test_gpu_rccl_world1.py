"""GPU: the collective path of the head-sharded driver on a ONE-rank RCCL group (backend "nccl"). A 1-GPU box cannot run
several ranks, but a 1-rank all-gather still goes through ProcessGroupNCCL: its own stream, async work handles, the
stream-ordering between the window launches and the gathers, bf16 views of one flat buffer — with the real kernels."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_overlapped_and_plain_all_gather_on_a_one_rank_rccl_group():
    import torch.distributed as dist
    import liteattention_amd as L
    from helpers import structured_qkv
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        B, S, H, D = 1, 2600, 4, 128
        q, k, v = [x.cuda() for x in structured_qkv(B, S, H, D, seed=21)]
        local = L.LiteAttention(threshold=-3.0, max_batch_size=B)
        plain = L.HeadShardedLiteAttention(num_heads=H, threshold=-3.0, max_batch_size=B, process_group=dist.group.WORLD,
                                           overlap_windows=1, _collective_at_world_1=True)
        over = L.HeadShardedLiteAttention(num_heads=H, threshold=-3.0, max_batch_size=B, process_group=dist.group.WORLD,
                                          overlap_windows=3, _collective_at_world_1=True)
        bm = L.get_tile_sizes(D, 2)[0]
        Qt, per = -(-S // bm), 1024 // bm
        over.q_windows = lambda q_: [(0, per), (per, per), (2 * per, Qt - 2 * per)]     # 1024 + 1024 + 552 rows
        for step in range(3):
            ref = local(q, k, v)
            g1 = plain(q, k, v)
            g2 = over(q, k, v)
            torch.cuda.synchronize()
            assert g1.shape == (1, B, S, H, D) and torch.equal(g1[0], ref)
            assert [tuple(b.shape) for b in g2] == [(1, B, 1024, H, D), (1, B, 1024, H, D), (1, B, S - 2048, H, D)]
            assert torch.equal(L.HeadShardedLiteAttention.to_bshd(g2), ref)
            assert torch.equal(plain.local._skip_list, local._skip_list) and torch.equal(over.local._skip_list, local._skip_list)
        assert local.get_skip_fraction() > 0.02
    finally:
        dist.destroy_process_group()
