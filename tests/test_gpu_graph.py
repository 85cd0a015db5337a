"""GPU: a denoising loop captured in a HIP graph. Nothing on the call path syncs with the host or depends on host state that changes
between replays EXCEPT the ping-pong phase of the lists (`LiteAttention._phase`, hopper/lite_attention.py:164-204): a captured call
replays with the read / write list pointers it was captured with, so the unit of capture is TWO consecutive calls (phases 0 and 1) -
one graph then advances the skip state by two denoising steps per replay, with no Python between the launches (la_fwd itself is a
memset of the ticket counters + one kernel on the capture stream; the library allocates nothing and keeps no state). The reference
has no counterpart (one op call per step from Python, flash_fwd_launch_template.h:359); this is the MI355X-side answer to launch-bound
inner loops: 40 layers x 2 phases of a Wan2.x step can sit in one graph."""
import pytest
import torch

from helpers import structured_qkv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_two_steps_per_replay_equal_the_eager_loop(dtype):
    import liteattention_amd as L
    B, S, H, D, thr, steps = 1, 2304, 4, 128, -3.0, 8
    cast = (lambda x: x.to(torch.float8_e4m3fn)) if dtype == "fp8" else (lambda x: x)
    data = [[cast(x.cuda()) for x in structured_qkv(B, S, H, D, seed=400, alpha=9.0 - 0.3 * t)] for t in range(steps)]
    # eager loop
    att_e = L.LiteAttention(threshold=thr, max_batch_size=B)
    outs_e = [att_e(*data[t], return_softmax_lse=True) for t in range(steps)]
    # graph: steps 0, 1 eagerly on the current stream (allocates the lists and the must-do row; they are the warm-up), then one graph of two calls
    att_g = L.LiteAttention(threshold=thr, max_batch_size=B)
    for t in (0, 1):
        o, lse = att_g(*data[t], return_softmax_lse=True)
        assert torch.equal(o, outs_e[t][0])
    static = [[torch.empty_like(x) for x in data[0]] for _ in range(2)]
    phase0 = att_g._phase
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o0, l0 = att_g(*static[0], return_softmax_lse=True)
        o1, l1 = att_g(*static[1], return_softmax_lse=True)
    assert att_g._phase == phase0                         # two calls: the host-side phase is back where the graph starts
    for t in range(2, steps, 2):
        for i in range(2):
            for buf, src in zip(static[i], data[t + i]):
                buf.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o0, outs_e[t][0]) and torch.equal(l0, outs_e[t][1]), t
        assert torch.equal(o1, outs_e[t + 1][0]) and torch.equal(l1, outs_e[t + 1][1]), t + 1
    n = int(att_e._skip_list[..., 0].max())
    live = torch.arange(n + 1, device="cuda") <= att_e._skip_list[..., 0:1]
    assert bool(((att_g._skip_list[..., : n + 1] == att_e._skip_list[..., : n + 1]) | ~live).all())     # same skip state after 8 steps
    assert att_g.get_skip_fraction() == att_e.get_skip_fraction() > 0.05


def test_list_growth_cannot_happen_inside_a_capture_and_preallocate_avoids_it():
    """the lists grow on demand when a larger batch appears - which replaces the tensor a captured graph points into. Growth
    inside a capture raises; ``preallocate`` sizes the lists for the largest batch first, and then a graph captured at batch 2 replays
    bit-identically to eager calls while batch-1 calls in between keep using the same (unmoved) lists."""
    import liteattention_amd as L
    S, H, D, thr = 1024, 2, 128, -3.0
    q1, k1, v1 = [x.cuda() for x in structured_qkv(1, S, H, D, seed=500, alpha=8.0)]
    q2, k2, v2 = [x.cuda() for x in structured_qkv(2, S, H, D, seed=501, alpha=8.0)]
    att = L.LiteAttention(threshold=thr, max_batch_size=2)
    att(q1, k1, v1); att(q1, k1, v1)                       # lists sized for batch 1
    assert att._skip_list.shape[1] == 1
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="preallocate"):
        with torch.cuda.graph(g):
            att(q2, k2, v2)                                # would have to grow inside the capture
    torch.cuda.synchronize()
    att = L.LiteAttention(threshold=thr, max_batch_size=2)
    att.preallocate(q1, v1)                                # max_batch_size sequences, before anything is captured
    assert att._skip_list.shape[1] == 2
    ptr = att._skip_list.data_ptr()
    ref = L.LiteAttention(threshold=thr, max_batch_size=2)
    e = [ref(q2, k2, v2), ref(q2, k2, v2), ref(q2, k2, v2), ref(q2, k2, v2)]
    att(q2, k2, v2); att(q2, k2, v2)                       # eager warm-up (must-do row), phases 0 and 1
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o0 = att(q2, k2, v2)
        o1 = att(q2, k2, v2)
    g.replay()
    torch.cuda.synchronize()
    assert att._skip_list.data_ptr() == ptr and torch.equal(o0, e[2]) and torch.equal(o1, e[3])


@pytest.mark.parametrize("D", [128, 192])
def test_dense_calls_on_the_ticket_queues_capture_and_replay(D, monkeypatch):
    """Round 5: dense launches of the hand-scheduled kernels run on persistent workgroups + ticket queues too (a 1 KiB workspace per
    call, its memset on the capture stream). Captured in a HIP graph and replayed on new inputs they equal the eager calls bit for bit -
    and the eager dense call equals the same call forced onto the static map (LA_FLAG_STATIC_SCHED): scheduling never changes results."""
    import liteattention_amd as L
    monkeypatch.delenv("LA_SCHED", raising=False)
    g = torch.Generator(device="cuda").manual_seed(77)
    S, H = 3000, 7                                               # 12 (24 at head_dim 192) q-tiles x 7 heads: more items than one round of stealing needs
    data = [[torch.randn(1, S, H, D, device="cuda", generator=g).bfloat16() for _ in range(3)] for _ in range(3)]
    eager = [L.flash_attn_func(*d, return_softmax_lse=True) for d in data]
    monkeypatch.setenv("LA_SCHED", "static")                     # the host layer's switch for LA_FLAG_STATIC_SCHED (read per call)
    static_map = [L.flash_attn_func(*d, return_softmax_lse=True) for d in data]
    monkeypatch.delenv("LA_SCHED")
    for (o, l), (os_, ls) in zip(eager, static_map):
        assert torch.equal(o, os_) and torch.equal(l, ls)
    buf = [torch.empty_like(x) for x in data[0]]
    for b, x in zip(buf, data[0]):
        b.copy_(x)
    L.flash_attn_func(*buf)                                      # warm-up outside the capture
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o_g, l_g = L.flash_attn_func(*buf, return_softmax_lse=True)
    for d, (o, l) in zip(data, eager):
        for b, x in zip(buf, d):
            b.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(o_g, o) and torch.equal(l_g, l)
