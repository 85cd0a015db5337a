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


def test_shim_import_names_and_reference_signatures():
    """compat_shims/ (opt-in on sys.path) exposes the operator surface under the reference's own module and function names with
    the reference's parameter lists (flash_attn/flash_attn_interface.py:998-1462, hopper/_internal/flash_attn_interface.py:487-682,
    flash_attn/flash_blocksparse_attn_interface.py:185-200). Host-side only: nothing is launched."""
    import importlib
    import inspect
    import os
    import sys
    shims = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "compat_shims")
    assert importlib.util.find_spec("flash_attn") is None                     # never shadowed by accident: opt-in directory
    sys.path.insert(0, shims)
    try:
        import flash_attn
        import flash_attn_interface as fa3
        from flash_attn import flash_blocksparse_attn_interface as bs
        names = lambda f: list(inspect.signature(f).parameters)            # noqa: E731
        fa2_tail = ["dropout_p", "softmax_scale", "causal", "window_size", "softcap", "alibi_slopes", "deterministic", "return_attn_probs"]
        assert names(flash_attn.flash_attn_func) == ["q", "k", "v"] + fa2_tail
        assert names(flash_attn.flash_attn_kvpacked_func) == ["q", "kv"] + fa2_tail
        assert names(flash_attn.flash_attn_qkvpacked_func) == ["qkv"] + fa2_tail
        assert names(flash_attn.flash_attn_varlen_func) == ["q", "k", "v", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k"] + fa2_tail + ["block_table"]
        assert names(flash_attn.flash_attn_varlen_kvpacked_func) == ["q", "kv", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k"] + fa2_tail
        assert names(flash_attn.flash_attn_varlen_qkvpacked_func) == ["qkv", "cu_seqlens", "max_seqlen"] + fa2_tail
        assert names(fa3.flash_attn_varlen_func) == ["q", "k", "v", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "seqused_q",
                                                     "seqused_k", "softmax_scale", "causal", "qv", "q_descale", "k_descale", "v_descale",
                                                     "window_size", "attention_chunk", "softcap", "num_splits", "pack_gqa", "deterministic",
                                                     "sm_margin"]
        assert names(fa3.flash_attn_func)[-5:] == ["attn_read_list", "attn_must_do_list", "attn_write_list", "thr", "return_softmax_lse"]
        assert names(bs.flash_blocksparse_attn_func) == ["qkv", "cu_seqlens", "blockmask", "dropout_p", "max_s", "softmax_scale", "causal",
                                                         "return_attn_probs", "convert_mask"]
        assert fa3.flash_attn_func is L.flash_attn_func and fa3.flash_attn_combine is L.flash_attn_combine
        for fn in (flash_attn.flash_attn_with_kvcache, fa3.flash_attn_with_kvcache, fa3.get_scheduler_metadata):
            with pytest.raises(NotImplementedError):
                fn()
        q = torch.zeros(64, 1, 128, dtype=torch.bfloat16)
        with pytest.raises(NotImplementedError):
            flash_attn.flash_attn_varlen_func(q, q, q, [0, 64], [0, 64], 64, 64, causal=True)
        with pytest.raises(NotImplementedError):
            fa3.flash_attn_varlen_func(q, q, q, [0, 64], [0, 64], 64, 64, seqused_k=torch.ones(1))
        with pytest.raises((RuntimeError, NotImplementedError)):                 # CPU tensors: no fallback behind the shim either
            flash_attn.flash_attn_varlen_func(q, q, q, [0, 64], [0, 64], 64, 64)
    finally:
        sys.path.remove(shims)
        for n in [m for m in sys.modules if m == "flash_attn" or m.startswith("flash_attn.") or m == "flash_attn_interface"]:
            del sys.modules[n]


def test_build_rejects_scratch_in_the_x64_kernels():
    """build.py::_check_no_scratch: the x64 kernels (an asm body that clobbers nearly the whole register file inside a C++ shell) must
    not spill — hipcc has placed such spill stores where EXEC is 0 (HISTORY.md section 3.1, 'A compiler hazard')."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("la_build_t", os.path.join(os.path.dirname(L.__file__), "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    ok = ("f.hip:1:1: remark: Function Name: _ZN2la27la_fwd_x64_kernelILb1EEEvNS_9FwdParamsE [-Rpass]\n"
          "f.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass]\n"
          "f.hip:1:1: remark: Function Name: _ZN2la21la_fwd_v2_kernelILi256ELb1EEEvNS_9FwdParamsE [-Rpass]\n"
          "f.hip:1:1: remark:     ScratchSize [bytes/lane]: 272 [-Rpass]\n")
    b._check_no_scratch(ok, ())                                                  # the hipcc-scheduled 128-row kernels may spill
    bad = ok.replace("ScratchSize [bytes/lane]: 0", "ScratchSize [bytes/lane]: 48")
    with pytest.raises(RuntimeError, match="must not spill"):
        b._check_no_scratch(bad, ())


def test_op_has_a_meta_kernel_for_fake_tensor_tracing():
    """torch.ops.lite_attention.fwd on meta tensors: shapes / dtypes of the reference op (flash_api.cpp:859, 887-892) without a launch,
    so that torch.compile / export can trace a pipeline through the opaque op."""
    import torch
    import liteattention_amd  # noqa: F401  (registers the op)
    q = torch.empty(2, 100, 6, 128, dtype=torch.bfloat16, device="meta")
    k = torch.empty(2, 130, 2, 128, dtype=torch.bfloat16, device="meta")
    out, lse, a, b = torch.ops.lite_attention.fwd(q, k, k)
    assert out.shape == q.shape and out.dtype == torch.bfloat16 and out.device.type == "meta"
    assert lse.shape == (2, 6, 100) and lse.dtype == torch.float32 and a.numel() == 0 and b.numel() == 0
    q8 = torch.empty(1, 64, 2, 128, dtype=torch.float8_e4m3fn, device="meta")
    assert torch.ops.lite_attention.fwd(q8, q8, q8)[0].dtype == torch.bfloat16
    qh = torch.empty(1, 64, 2, 96, dtype=torch.float16, device="meta")
    assert torch.ops.lite_attention.fwd(qh, qh, qh)[0].dtype == torch.float16
    qp = torch.empty(300, 4, 128, dtype=torch.bfloat16, device="meta")
    cu = torch.empty(4, dtype=torch.int32, device="meta")
    o, l, _, _ = torch.ops.lite_attention.fwd(qp, qp, qp, None, None, None, None, cu, cu, None, None, None, 120, 120)
    assert o.shape == (300, 4, 128) and l.shape == (4, 300)


@pytest.mark.parametrize("B,H", [(3, 3), (2, 5), (4, 1)])
@pytest.mark.parametrize("rank", [2, 3, 4])
def test_host_blockmask_k_tiles_valid_is_per_batch(B, H, rank):
    """on the host path a 4-D mask [B, H, q, k] used to line `k_tiles_valid` [B] up with the HEADS axis (wrong lists when
    B == H, a broadcast error otherwise). Row-by-row reference: the rows of `blockmask_to_rows` on the mask with the columns beyond
    batch b's valid count cleared."""
    from liteattention_amd.compat import blockmask_to_lists
    g = torch.Generator().manual_seed(5 + 10 * B + H)
    qt, kt = 4, 9
    shape = {2: (qt, kt), 3: (B, qt, kt), 4: (B, H, qt, kt)}[rank]
    mask = torch.rand(shape, generator=g) < 0.5
    mask[..., 0] = True                                            # every row keeps tile 0: valid under every k_tiles_valid >= 1
    kv = torch.randint(1, kt + 1, (B,), generator=g)
    got = blockmask_to_lists(mask, k_tiles_valid=kv)
    full = mask.expand(B, qt, kt) if rank == 2 else mask
    assert got.shape[0] == B and got.shape[-2:] == (qt, kt + 1)
    for b in range(B):
        for h in range(H if rank == 4 else 1):
            m2 = (full[b, h] if rank == 4 else full[b]).clone()
            m2[:, int(kv[b]):] = False
            want = blockmask_to_rows(m2)
            rows = got[b, h] if rank == 4 else got[b]
            for r, w in zip(rows.tolist(), want):
                assert r[: len(w)] == w and not any(r[len(w):]), (b, h, r, w)
    if rank >= 3:
        with pytest.raises(ValueError):
            blockmask_to_lists(mask, k_tiles_valid=torch.ones(B + 1, dtype=torch.int64))
