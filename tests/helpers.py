"""Shared test helpers: seeded inputs (same recipe as tests/golden/make_golden.py) and golden loading."""
import json
import math
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DENSE_CASES = ["cfg0_fp32_s2048_d64", "bf16_b2_s333_h3_d128", "bf16_s512_h2_d128", "bf16_sq113_sk203_h2_d128"]
FP16_CASES = ["fp16_b2_s333_h3_d128", "fp16_sq130_sk517_h2_d64"]
FP8_CASES = ["fp8_b2_s333_h3_d128", "fp8_sq200_sk777_h2_d128", "fp8_sq130_sk517_h2_d64"]
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


def fp8_lse_tol_vs_exact():
    """Bound on |LSE - EXACT LSE| (the oracle with un-rounded P) for the fp8 kernel, by the form of P that is selected
    (include/lite_attention_amd.h):
    LA_FP8_P=encoded (LA_FLAG_FP8_ENCODED_P) - the block-scaled log-linear byte encoding of P, row sums of the ENCODED P from the matrix pipe: every P~ / P lies in
      [0.920, 1.065] (tests/test_oracle.py scans it), so |ln(sum P~ / sum P)| <= -ln 0.920 = 0.083, reached only by rows of one or two
      comparable keys; on long rows the noise averages out and a bias of about -3e-4 remains;
    LA_FP8_P=mfma_rowsum (LA_FLAG_FP8_MFMA_ROWSUM) - v_exp_f32 + hardware e4m3 rounding, row sums of the ROUNDED P: every P~ within 2^-4 of its
      P, ln(1 + 2^-4) = 0.0606 (1e-2 typical for a few keys; about -7e-4 of bias on long rows);
    default (the reference's arithmetic since round 6) - fp32 sums of the un-rounded P (softmax.h:275-296): 1e-3, the bf16 bound."""
    form = fp8_form()
    if form == "reference":
        return 1e-3
    if form == "mfma_rowsum":
        return math.log1p(2.0 ** -4) + 1e-3
    return -math.log(0.920) + 1e-3


def fp8_form():
    """The fp8 form of P the process runs with (liteattention_amd._cabi.default_flags): "reference" (default), "mfma_rowsum", "encoded"."""
    return os.environ.get("LA_FP8_P", "") or "reference"


def fp8_lse_tol():
    """Bound on |LSE - oracle IN THE SAME FORM of P| (``fp8_p_round()``). The reference forms: as ``fp8_lse_tol_vs_exact``. The encoded form:
    since round 5 the oracle encodes P~ on the kernel's own grid (relative to the lazy reference maximum), so the two agree to ~1e-4 on
    all but the rows where a score sits on a byte boundary and the last bits of S (MFMA accumulation order vs the CPU's) decide it; one
    byte of a row's dominant key is a factor of up to 1.125 (e4m3 mantissa step at M = 0) = 0.118 in the LSE. So: 0.118 + 1e-3 for the
    worst row - and ``fp8_rows_off_grid`` holds the NUMBER of such rows down, which is what says the grids are the same."""
    if fp8_form() != "encoded":
        return fp8_lse_tol_vs_exact()
    return math.log(1.125) + 1e-3


def fp8_rows_off_grid(lse, lse_same_form, atol=0.01):
    """Fraction of rows whose LSE differs from the same-form oracle's by more than `atol` (a flipped byte of a significant key).
    With the oracle on the kernel's grid this is a fraction of a percent (rounds 3-4, another grid: ~60 %)."""
    d = (lse - lse_same_form).abs()
    d = d[torch.isfinite(d)]
    return float((d > atol).float().mean().item()) if d.numel() else 0.0


def fp8_p_round():
    """The oracle's `p_round` that restates the form of P the fp8 kernel is running with (see fp8_lse_tol): "fp8_lin" for the default
    log-linear byte encoding (LA_FP8_P=encoded), "fp8" (the reference's e4m3 rounding of exp2) for the default and LA_FP8_P=mfma_rowsum."""
    return "fp8_lin" if fp8_form() == "encoded" else "fp8"


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


def fragmented_qkv(B, S, H, D, seed, step=0, steps=6, alpha=10.0, block_n=64, dtype=torch.bfloat16, Sk=None):
    """Inputs whose QK-Skip lists FRAGMENT: key tiles (of `block_n` keys) are hot every 3rd or 4th tile (per-head pattern) and cold between, cold
    tiles score `1 - g` below the hot ones (g per tile, fixed by `seed`), so a negative threshold flags most cold pairs and every
    flagged pair splits the list (the first flagged tile of a run stays as the range end): about Kt / 3.5 ranges per
    row, i.e. more than the 64 ranges one pass of the wave-parallel list expansion handles once Kt > 200. Some hot tiles are only
    lukewarm and some cold ones nearly hot, and the per-step noise shrinks (`step` / `steps`), so lists keep changing over steps.
    Returns bf16-representable (q, k, v) of shape (B, S|Sk, H, D) on the CPU."""
    Sk = S if Sk is None else Sk
    g = torch.Generator().manual_seed(seed)
    kt = -(-Sk // block_n)
    u = torch.randn(B, 1, H, D, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    pos = torch.cumsum(torch.randint(3, 5, (B, H, kt), generator=g), dim=-1) - 3   # a hot tile every 3 or 4 tiles, per head
    hot = torch.zeros(B, H, kt + 4, dtype=torch.bool).scatter_(2, pos.clamp(max=kt + 3), True)[..., :kt]
    gain_hot = 0.62 + 0.38 * torch.rand(B, H, kt, generator=g)
    gain_cold = 0.45 * torch.rand(B, H, kt, generator=g) ** 0.5
    gain = torch.where(hot, gain_hot, gain_cold)
    gain[..., kt - 1] = 1.0                                                       # the first walked tile sets the running max
    gain_keys = gain.permute(0, 2, 1).repeat_interleave(block_n, dim=1)[:, :Sk]   # (B, Sk, H)
    q0 = alpha * u + torch.randn(B, S, H, D, generator=g)
    k0 = alpha * gain_keys[..., None] * u + torch.randn(B, Sk, H, D, generator=g)
    v0 = torch.randn(B, Sk, H, D, generator=g)
    s = 0.6 * (1.0 - step / max(1, steps))                                        # per-step perturbation, shrinking
    gs = torch.Generator().manual_seed(seed * 1000 + 17 + step)
    q = q0 + s * torch.randn(q0.shape, generator=gs)
    k = k0 + s * torch.randn(k0.shape, generator=gs)
    return q.to(dtype), k.to(dtype), v0.to(dtype)
