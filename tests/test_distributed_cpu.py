"""CPU, world_size 2, gloo: the head-sharding + all-gather path of liteattention_amd.parallel.
The device op cannot run here (no CPU fallback in the product), so the attention is replaced through the
documented test seam by the eager oracle; what is under test is partitioning, state locality and the collective."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from liteattention_amd.parallel import HeadShardedLiteAttention, head_range
        from oracle import oracle as orc

        calls = []

        def stand_in(q, k, v, scale=None, **kw):
            calls.append(q.shape)
            return orc.attention_dense_ref(q, k, v, softmax_scale=scale)[0]

        H = 6
        g = torch.Generator().manual_seed(0)           # same full tensors on every rank
        q, k, v = [torch.randn(2, 96, H, 32, generator=g) for _ in range(3)]
        att = HeadShardedLiteAttention(num_heads=H, max_batch_size=2, process_group=dist.group.WORLD,
                                       attention_fn=stand_in)
        assert (att.h0, att.h1) == head_range(H, world, rank) == (rank * 3, rank * 3 + 3)
        gathered = att(att.shard(q), att.shard(k), att.shard(v))
        assert gathered.shape == (world, 2, 96, 3, 32) and calls == [(2, 96, 3, 32)]
        full = HeadShardedLiteAttention.to_bshd(gathered)
        ref = orc.attention_dense_ref(q, k, v)[0]
        assert torch.allclose(full, ref, atol=1e-6), (full - ref).abs().max()
        local = att(att.shard(q), att.shard(k), att.shard(v), gather=False)
        assert torch.equal(local, ref[:, :, att.h0:att.h1])
        # ---- overlapped form: q-tile windows, one async all-gather per window, row-chunked result
        from liteattention_amd.parallel import plan_q_windows
        hooks = []

        def windowed_stand_in(q, k, v, windows, hook, scale=None, **kw):
            out = orc.attention_dense_ref(q, k, v, softmax_scale=scale)[0]
            for i, (t0, n) in enumerate(windows):      # the device op finalises rows window by window, in stream order
                hooks.append((i, t0 * 16, min(q.shape[1], (t0 + n) * 16)))
                hook(i, out, t0 * 16, min(q.shape[1], (t0 + n) * 16))
            return out

        att2 = HeadShardedLiteAttention(num_heads=H, max_batch_size=2, process_group=dist.group.WORLD,
                                        overlap_windows=3, windowed_attention_fn=windowed_stand_in, q_tile_rows=16)
        att2.q_windows = lambda q: [(0, 2), (2, 3), (5, 1)]              # 6 q-tiles of 16 rows, uneven windows
        blocks = att2(att2.shard(q), att2.shard(k), att2.shard(v))
        assert [tuple(b.shape) for b in blocks] == [(world, 2, 32, 3, 32), (world, 2, 48, 3, 32), (world, 2, 16, 3, 32)]
        assert hooks == [(0, 0, 32), (1, 32, 80), (2, 80, 96)]
        assert torch.allclose(HeadShardedLiteAttention.to_bshd(blocks), ref, atol=1e-6)
        # window planning: whole rounds of 256 workgroups, last window takes the remainder
        assert plan_q_windows(296, 5, 3) == [(0, 102), (102, 102), (204, 92)]       # 8 GPUs: 5 heads x 296 q-tiles
        assert plan_q_windows(296, 20, 3) == [(0, 102), (102, 102), (204, 92)]      # 2 GPUs: 8 rounds of 256 = 102 q-tiles
        assert plan_q_windows(296, 40, 1) == [(0, 296)] and plan_q_windows(1, 40, 4) == [(0, 1)]
        for qt, wg, n in [(296, 5, 3), (296, 10, 4), (591, 5, 3), (7, 3, 5), (128, 40, 2)]:
            w = plan_q_windows(qt, wg, n)
            assert w[0][0] == 0 and all(a + c == b for (a, c), (b, _) in zip(w[:-1], w[1:])) and sum(c for _, c in w) == qt
        # ---- Ulysses: sequence shards in and out, two all-to-alls, heads sharded inside
        from liteattention_amd.parallel import RingSeqParallelLiteAttention, UlyssesLiteAttention
        Sl = q.shape[1] // world
        rows = slice(rank * Sl, (rank + 1) * Sl)
        calls.clear()
        uly = UlyssesLiteAttention(num_heads=H, max_batch_size=2, process_group=dist.group.WORLD, attention_fn=stand_in)
        qh = uly.seq_to_head(q[:, rows].contiguous())
        assert torch.equal(qh, q[:, :, rank * 3: rank * 3 + 3])             # all rows of my heads
        assert torch.equal(uly.head_to_seq(qh), q[:, rows])                  # and back
        o_u = uly(q[:, rows].contiguous(), k[:, rows].contiguous(), v[:, rows].contiguous())
        assert calls == [(2, 96, 3, 32)] and o_u.shape == (2, Sl, H, 32)
        assert torch.allclose(o_u, ref[:, rows], atol=1e-6)
        # ---- ring: K/V shards travel, state j for (my Q x shard j), LSE merge of the partial results
        used = []

        def partial_attention(qq, kk, vv, split_idx, scale=None):
            used.append(split_idx)
            o, l = orc.attention_dense_ref(qq, kk, vv, softmax_scale=scale)
            return o, l

        def combine(outs, lses):                                             # (G,B,Sl,H,D), (G,B,H,Sl)
            return orc.attention_combine_ref(outs, lses.transpose(2, 3))[0]

        ring = RingSeqParallelLiteAttention(max_batch_size=2, process_group=dist.group.WORLD,
                                            attention_fn=partial_attention, combine_fn=combine)
        o_r = ring(q[:, rows].contiguous(), k[:, rows].contiguous(), v[:, rows].contiguous())
        assert used == [rank, (rank - 1) % world]                            # own shard first, then the previous rank's
        assert torch.allclose(o_r, ref[:, rows], atol=1e-5), (o_r - ref[:, rows]).abs().max()
        assert len(ring.states.lite_attention) == world
        with pytest.raises(AssertionError):
            att(q, k, v)                               # full tensors are not a local shard
        with pytest.raises(ValueError):
            head_range(5, 2, 0)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_head_sharded_all_gather_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


# ------------------------------------------------------------------------------------------ bench.py's N > 1 control flow, 4 ranks
def _bench_worker(rank, world, port, tmpdir):
    """the trial step of the overlapped all-gather, the agreement all-reduce, the fallback path (one rank is
    made to fail) and the multi_gpu record - the module-level functions bench.py's main() calls - with MORE than two ranks. The
    attention is a stand-in (the device op has no CPU form); heads 40 -> 10 per rank as SURVEY.md 8(e) partitions them."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from liteattention_amd.parallel import HeadShardedLiteAttention
        from oracle import oracle as orc
        dev = torch.device("cpu")
        H, B, S, D = 40, 1, 64, 16
        g = torch.Generator().manual_seed(0)
        q, k, v = [torch.randn(B, S, H, D, generator=g) for _ in range(3)]
        fail_on = {"rank": 2, "armed": True}
        windowed_calls, plain_calls = [], []

        def windowed_stand_in(qq, kk, vv, windows, hook, scale=None, **kw):
            windowed_calls.append(len(windows))
            if fail_on["armed"] and rank == fail_on["rank"]:
                raise RuntimeError("injected: this rank cannot run the overlapped form")
            out = orc.attention_dense_ref(qq, kk, vv, softmax_scale=scale)[0]
            for i, (t0, n) in enumerate(windows):
                hook(i, out, t0 * 16, min(qq.shape[1], (t0 + n) * 16))
            return out

        def stand_in(qq, kk, vv, scale=None, **kw):
            plain_calls.append(tuple(qq.shape))
            return orc.attention_dense_ref(qq, kk, vv, softmax_scale=scale)[0]

        def make(overlap):
            a = HeadShardedLiteAttention(num_heads=H, max_batch_size=B, process_group=dist.group.WORLD, overlap_windows=overlap,
                                         attention_fn=stand_in, windowed_attention_fn=windowed_stand_in, q_tile_rows=16, slots=20)
            return a, (a.shard(q), a.shard(k), a.shard(v))

        assert bench.require_world(dist, 4) == 4
        with pytest.raises(SystemExit):
            bench.require_world(dist, 8)                      # the launcher made 4 ranks: a line for 8 would be a lie
        with pytest.raises(SystemExit):
            bench.require_world(None, 2)
        barrier = dist.barrier

        # 1) a rank fails the trial step -> EVERY rank falls back; the fallback (kernel, then one all-gather) is what gets timed
        att, qkv = make(3)
        assert (att.h0, att.h1) == (10 * rank, 10 * rank + 10) and len(att.q_windows(qkv[0])) >= 2
        note = bench.agree_on_overlapped_form(att, qkv, dist, dev, lambda: None)
        assert att.overlap_windows == 1 and note is not None
        assert ("injected" in note) == (rank == 2) and ("another rank" in note) == (rank != 2)
        n_plain = len(plain_calls)
        step_s, kern_s = bench.timed_steps(att, qkv, 3, 1, barrier, dist, dev, bench.HostEvent)
        assert len(plain_calls) - n_plain == 4 and step_s > 0 and kern_s > 0 and kern_s <= step_s * 1.5
        rec = bench.multi_gpu_record(dist, dev, kern_s, att, qkv[0])
        assert rec["rccl_world_size"] == 4 and rec["backend"] == "gloo" and len(rec["kernel_ms_per_rank"]) == 4
        assert rec["heads_per_rank"] == [[0, 10], [10, 20], [20, 30], [30, 40]]
        assert rec["output_shard_bytes"] == B * S * 10 * D * 2 and rec["bytes_received_per_rank_per_step"] == 3 * rec["output_shard_bytes"]
        assert rec["overlapped_form_kept"] is False and rec["overlap_windows"] == 1
        full = HeadShardedLiteAttention.to_bshd(att(*qkv))
        ref = orc.attention_dense_ref(q, k, v)[0]
        assert torch.allclose(full, ref, atol=1e-6)

        # 2) nobody fails -> the overlapped form is kept and timed; same numbers
        fail_on["armed"] = False
        att, qkv = make(3)
        assert bench.agree_on_overlapped_form(att, qkv, dist, dev, lambda: None) is None and att.overlap_windows == 3
        n_w = len(windowed_calls)
        step_s, kern_s = bench.timed_steps(att, qkv, 2, 1, barrier, dist, dev, bench.HostEvent)
        assert len(windowed_calls) - n_w == 3 and n_w == 3      # before the timed loop: rank 2's failed preflight, this preflight, the trial step
        rec = bench.multi_gpu_record(dist, dev, kern_s, att, qkv[0])
        assert rec["overlapped_form_kept"] is True and rec["overlap_windows"] == len(att.q_windows(qkv[0])) >= 2
        assert torch.allclose(HeadShardedLiteAttention.to_bshd(att(*qkv)), ref, atol=1e-6)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_bench_multi_gpu_control_flow_world4(tmp_path):
    world = 4
    port = _free_port()
    mp.spawn(_bench_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1", "ok2", "ok3"]
