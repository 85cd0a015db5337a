"""Shared test helpers: seeded inputs (same recipe as tests/golden/make_golden.py) and golden loading."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DENSE_CASES = ["cfg0_fp32_s2048_d64", "bf16_b2_s333_h3_d128", "bf16_s512_h2_d128", "bf16_sq113_sk203_h2_d128"]
FP16_CASES = ["fp16_b2_s333_h3_d128", "fp16_sq130_sk517_h2_d64"]
FP8_CASES = ["fp8_b2_s333_h3_d128", "fp8_sq200_sk777_h2_d128"]
GQA_CASES = ["gqa_bf16_b2_s200_h6_hk2_d128", "mqa_bf16_sq130_sk517_h4_hk1_d64"]     # nheads_k < nheads
GQA_FP8_CASES = ["gqa_fp8_b1_s260_h4_hk2_d128"]


def dense_inputs(seed, B, Sq, Sk, H, D, dtype, Hk=None):
    """randn fp32 -> dtype -> fp32 (hopper/tests/test_flash_attn.py:204-210), CPU generator. Hk = K/V heads."""
    Hk = H if Hk is None else Hk
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, Sq, H, D, generator=g).to(dtype).float()
    k = torch.randn(B, Sk, Hk, D, generator=g).to(dtype).float()
    v = torch.randn(B, Sk, Hk, D, generator=g).to(dtype).float()
    return q, k, v


def load_dense_case(name):
    z = np.load(os.path.join(GOLDEN, f"dense_{name}.npz"))
    seed, B, Sq, Sk, H, D = [int(x) for x in z["meta"][:6]]
    Hk = int(z["meta"][6]) if len(z["meta"]) > 6 else H
    dtype = getattr(torch, str(z["dtype"]))
    q, k, v = dense_inputs(seed, B, Sq, Sk, H, D, dtype, Hk)
    chk = q.double().sum().item() + 2 * k.double().sum().item() + 3 * v.double().sum().item()
    assert abs(chk - float(z["input_checksum"])) < 1e-6, "torch CPU generator drifted: regenerate tests/golden"
    case = {"q": q, "k": k, "v": v, "dtype": dtype, "out_ref": torch.from_numpy(z["out_ref"]),
            "lse_ref": torch.from_numpy(z["lse_ref"]), "pt_maxerr": float(z["pt_maxerr"]), "D": D}
    for name in ("q_descale", "k_descale", "v_descale"):
        if name in z.files:
            case[name] = torch.from_numpy(z[name])
    return case


def ref_tolerance(out_ref, pt_maxerr):
    """Reference rule, hopper/tests/test_flash_attn.py:283,296."""
    fwd_atol = 2 * (out_ref + 0.3 - 0.3 - out_ref).abs().max().item()
    return 2 * pt_maxerr + fwd_atol


def host_golden():
    with open(os.path.join(GOLDEN, "host_golden.json")) as f:
        return json.load(f)


def structured_qkv(B, S, H, D, seed, alpha=10.0, frames=8, dtype=torch.bfloat16):
    """Inputs with structured (frame-clustered) attention so that negative thresholds produce real
    sparsity (iid randn gives ~0). Small-scale version of the generator of SURVEY.md §8(d)."""
    g = torch.Generator().manual_seed(seed)
    per = -(-S // frames)
    fidx = torch.arange(S) // per
    out = []
    u = torch.randn(B, frames, H, D, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    for which in range(2):
        x = alpha * u[:, fidx] + torch.randn(B, S, H, D, generator=g)
        out.append(x.to(dtype))
    out.append(torch.randn(B, S, H, D, generator=g).to(dtype))
    return out
