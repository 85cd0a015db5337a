"""Generates tests/golden/* by IMPORTING the reference's Python (run in the build container only).

    python tests/golden/make_golden.py          # needs /root/reference; never runs on the GPU box

Nothing from the reference is copied: only inputs/outputs (data) are stored.
  host_golden.json   G1 get_MN table, G2 init_skip_list rows, G3 _expand_must_do_list rows,
                     G4 host call trace (ping-pong phases, thr forwarding, re-init, reset)
                     <- /root/reference/hopper/lite_attention.py
  dense_*.npz        G5 outputs of attention_ref (fp32-upcast `out_ref`, same-dtype reordered `out_pt`
                     error) and the logsumexp of test_lite_attention.py:67-77
                     <- /root/reference/hopper/tests/test_util.py:226-348
Inputs are regenerated from the seed on both sides (torch CPU generator); a checksum guards drift.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.modules["lite_attention._C"] = types.ModuleType("lite_attention._C")
    spec = importlib.util.spec_from_file_location(
        "lite_attention", f"{REF}/hopper/__init__.py", submodule_search_locations=[f"{REF}/hopper"])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lite_attention"] = mod
    spec.loader.exec_module(mod)
    sys.path += [f"{REF}/hopper/utils", f"{REF}/hopper/tests"]
    import test_util  # noqa
    return mod, test_util


def dense_inputs(seed, B, Sq, Sk, H, D, dtype, Hk=None):
    """Input recipe of hopper/tests/test_flash_attn.py:204-210: randn fp32 -> dtype -> fp32. Hk = K/V heads (GQA)."""
    Hk = H if Hk is None else Hk
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Sq, H, D, generator=g).to(dtype).float()
    k = torch.randn(B, Sk, Hk, D, generator=g).to(dtype).float()
    v = torch.randn(B, Sk, Hk, D, generator=g).to(dtype).float()
    return q, k, v


DENSE_CASES = [  # name, seed, B, Sq, Sk, H, D, dtype
    ("cfg0_fp32_s2048_d64", 0, 1, 2048, 2048, 1, 64, "float32"),     # BASELINE.json configs[0]
    ("bf16_b2_s333_h3_d128", 1, 2, 333, 333, 3, 128, "bfloat16"),
    ("bf16_s512_h2_d128", 2, 1, 512, 512, 2, 128, "bfloat16"),
    ("bf16_sq113_sk203_h2_d128", 3, 1, 113, 203, 2, 128, "bfloat16"),
    ("fp16_b2_s333_h3_d128", 4, 2, 333, 333, 3, 128, "float16"),     # the reference's third dtype (flash_api.cpp:715)
    ("fp16_sq130_sk517_h2_d64", 5, 1, 130, 517, 2, 64, "float16"),
]


def main():
    mod, test_util = load_reference()
    LA = mod.LiteAttention
    host = {}

    # G1: get_MN table
    host["get_MN"] = [[d, es, vc, list(LA.get_MN(d, es, vc))]
                      for es in (2, 1) for d in (32, 64, 96, 128, 192, 256) for vc in (False, True)]

    # G2: init_skip_list
    g2 = []
    for (S, D, dt) in [(1000, 128, torch.bfloat16), (5000, 64, torch.bfloat16), (75600, 128, torch.bfloat16),
                       (75600, 128, torch.float8_e4m3fn)]:
        sl = LA.init_skip_list(1, S, 2, D, False, dt, "cpu")
        g2.append({"S": S, "D": D, "elem": dt.itemsize, "shape": list(sl.shape),
                   "row_head": sl[0, 0, 0, 0, :4].tolist(), "sum": int(sl.to(torch.int64).sum()),
                   "all_rows_equal": bool((sl == sl[0, 0, 0, 0]).all())})
    host["init_skip_list"] = g2

    # G3: _expand_must_do_list (kTileN = 176 for bf16 d=128)
    g3 = []
    for S, lst in [(5000, [0, 0]), (5000, [4999, 0]), (5000, [900, 300]), (100 * 176, [80, 60, 45, 40, 12, 2]),
                   (75600, [70000, 65000, 1000, 10])]:
        q = torch.zeros(1, S, 1, 128, dtype=torch.bfloat16)
        Kt = -(-S // 176)
        out = LA._expand_must_do_list(list(lst), (1, 1, 2, Kt + 1), q, q)
        g3.append({"S": S, "list": lst, "k_tile": 176, "width": Kt + 1, "shape": list(out.shape),
                   "row": out[0, 0, 0, :12].tolist(), "rows_equal": bool((out == out[0, 0, 0]).all()),
                   "tail_zero": bool((out[0, 0, 0, 12:] == 0).all())})
    host["expand_must_do_list"] = g3

    # G4: host call trace with the op replaced by a recorder
    trace = []
    holder = {}

    def recorder(q, k, v, softmax_scale=None, attn_read_list=None, attn_must_do_list=None, attn_write_list=None,
                 thr=None, return_softmax_lse=False, **kw):
        sl = holder["obj"]._skip_list
        base = sl.data_ptr()
        per = sl[0].numel() * 4
        trace.append({"read": (attn_read_list.data_ptr() - base) // per,
                      "write": (attn_write_list.data_ptr() - base) // per, "thr": thr,
                      "list_shape": list(attn_read_list.shape), "must_do_shape": list(attn_must_do_list.shape),
                      "must_do_head": attn_must_do_list[0, 0, 0, :3].tolist(), "scale": softmax_scale,
                      "phase_after": holder["obj"]._phase})
        return torch.zeros_like(q)

    mod.lite_attention.flash_attn_func = recorder
    att = LA(enable_skipping=True, threshold=-7.5, max_batch_size=2)
    holder["obj"] = att
    q1 = torch.zeros(1, 1000, 2, 128, dtype=torch.bfloat16)
    q2 = torch.zeros(2, 700, 2, 128, dtype=torch.bfloat16)
    steps = []
    att(q1, q1, q1); steps.append("call S=1000")
    att(q1, q1, q1); steps.append("call S=1000")
    att(q1, q1, q1, scale=0.25); steps.append("call S=1000 scale=0.25")
    att(q2, q2, q2); steps.append("call S=700 B=2 (re-init)")
    att(q2, q2, q2); steps.append("call S=700 B=2")
    att.reset_skip_state(); steps.append("reset")
    att.set_threshold(-2.0)
    att(q2, q2, q2); steps.append("call S=700 B=2 thr=-2")
    host["call_trace"] = {"ctor": {"threshold": -7.5, "max_batch_size": 2}, "steps": steps, "trace": trace,
                          "k_tile": 176, "q_tile": 128}
    # threshold guard
    try:
        LA(threshold=1.0)
        host["threshold_guard"] = "no error"
    except ValueError as e:
        host["threshold_guard"] = str(e)

    with open(os.path.join(HERE, "host_golden.json"), "w") as f:
        json.dump(host, f, indent=1)

    # G5: dense numerics
    for name, seed, B, Sq, Sk, H, D, dt in DENSE_CASES:
        dtype = getattr(torch, dt)
        q, k, v = dense_inputs(seed, B, Sq, Sk, H, D, dtype)
        out_ref, _ = test_util.attention_ref(q, k, v, None, None)
        qd, kd, vd = q.to(dtype), k.to(dtype), v.to(dtype)
        out_pt, _ = test_util.attention_ref(qd, kd, vd, None, None, upcast=False, reorder_ops=True)
        scores = torch.matmul(q.transpose(1, 2), k.transpose(1, 2).transpose(-2, -1)) * (1.0 / D ** 0.5)
        lse_ref = torch.logsumexp(scores, dim=-1)
        np.savez_compressed(
            os.path.join(HERE, f"dense_{name}.npz"),
            out_ref=out_ref.numpy().astype(np.float32), lse_ref=lse_ref.numpy().astype(np.float32),
            pt_maxerr=np.float32((out_pt.float() - out_ref).abs().max().item()),
            input_checksum=np.float64(q.double().sum().item() + 2 * k.double().sum().item() + 3 * v.double().sum().item()),
            meta=np.array([seed, B, Sq, Sk, H, D]), dtype=np.array(dt))
        print(name, "pt_maxerr", (out_pt.float() - out_ref).abs().max().item())
    # G5-fp8: e4m3 inputs with per-(batch, head) descales (hopper/tests/test_flash_attn.py:204-219, 253)
    for name, seed, B, Sq, Sk, H, D in [("fp8_b2_s333_h3_d128", 11, 2, 333, 333, 3, 128),
                                        ("fp8_sq200_sk777_h2_d128", 12, 1, 200, 777, 2, 128),
                                        ("fp8_sq130_sk517_h2_d64", 13, 1, 130, 517, 2, 64)]:      # round 6: the native head_dim-64 fp8 body
        q, k, v = dense_inputs(seed, B, Sq, Sk, H, D, torch.float8_e4m3fn)
        g = torch.Generator().manual_seed(1000 + seed)
        qd, kd, vd = [torch.rand(B, H, generator=g) * 2 for _ in range(3)]
        out_ref, _ = test_util.attention_ref(q, k, v, None, None, q_descale=qd, k_descale=kd, v_descale=vd)
        out_pt, _ = test_util.attention_ref(q, k, v, None, None, q_descale=qd, k_descale=kd, v_descale=vd,
                                            upcast=False, reorder_ops=True, intermediate_dtype=torch.float8_e4m3fn)
        scores = torch.einsum("bthd,bshd->bhts", q, k) * (qd * kd)[:, :, None, None] * (1.0 / D ** 0.5)
        lse_ref = torch.logsumexp(scores, dim=-1)
        np.savez_compressed(
            os.path.join(HERE, f"dense_{name}.npz"),
            out_ref=out_ref.numpy().astype(np.float32), lse_ref=lse_ref.numpy().astype(np.float32),
            pt_maxerr=np.float32((out_pt.float() - out_ref).abs().max().item()),
            input_checksum=np.float64(q.double().sum().item() + 2 * k.double().sum().item() + 3 * v.double().sum().item()),
            meta=np.array([seed, B, Sq, Sk, H, D]), dtype=np.array("float8_e4m3fn"),
            q_descale=qd.numpy(), k_descale=kd.numpy(), v_descale=vd.numpy())
        print(name, "pt_maxerr", (out_pt.float() - out_ref).abs().max().item())
    # G5-GQA: nheads_k < nheads (attention_ref repeats the K/V heads, test_util.py:283-284; the fp8 descales are per
    # K/V head, test_flash_attn.py:219). meta carries Hk as a 7th entry.
    for name, seed, B, Sq, Sk, H, Hk, D, dt in [("gqa_bf16_b2_s200_h6_hk2_d128", 21, 2, 200, 200, 6, 2, 128, "bfloat16"),
                                                ("mqa_bf16_sq130_sk517_h4_hk1_d64", 22, 1, 130, 517, 4, 1, 64, "bfloat16"),
                                                ("gqa_fp8_b1_s260_h4_hk2_d128", 23, 1, 260, 260, 4, 2, 128, "float8_e4m3fn")]:
        dtype = getattr(torch, dt)
        q, k, v = dense_inputs(seed, B, Sq, Sk, H, D, dtype, Hk)
        extra = {}
        if dt == "float8_e4m3fn":
            g = torch.Generator().manual_seed(1000 + seed)
            qd, kd, vd = [torch.rand(B, Hk, generator=g) * 2 for _ in range(3)]
            out_ref, _ = test_util.attention_ref(q, k, v, None, None, q_descale=qd, k_descale=kd, v_descale=vd)
            out_pt, _ = test_util.attention_ref(q, k, v, None, None, q_descale=qd, k_descale=kd, v_descale=vd,
                                                upcast=False, reorder_ops=True, intermediate_dtype=torch.float8_e4m3fn)
            sc = (qd * kd).repeat_interleave(H // Hk, dim=1)[:, :, None, None]
            extra = dict(q_descale=qd.numpy(), k_descale=kd.numpy(), v_descale=vd.numpy())
        else:
            out_ref, _ = test_util.attention_ref(q, k, v, None, None)
            out_pt, _ = test_util.attention_ref(q.to(dtype), k.to(dtype), v.to(dtype), None, None, upcast=False,
                                                reorder_ops=True)
            sc = 1.0
        kr = k.repeat_interleave(H // Hk, dim=2)
        scores = torch.einsum("bthd,bshd->bhts", q, kr) * sc * (1.0 / D ** 0.5)
        lse_ref = torch.logsumexp(scores, dim=-1)
        np.savez_compressed(
            os.path.join(HERE, f"dense_{name}.npz"),
            out_ref=out_ref.numpy().astype(np.float32), lse_ref=lse_ref.numpy().astype(np.float32),
            pt_maxerr=np.float32((out_pt.float() - out_ref).abs().max().item()),
            input_checksum=np.float64(q.double().sum().item() + 2 * k.double().sum().item() + 3 * v.double().sum().item()),
            meta=np.array([seed, B, Sq, Sk, H, D, Hk]), dtype=np.array(dt), **extra)
        print(name, "pt_maxerr", (out_pt.float() - out_ref).abs().max().item())
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
