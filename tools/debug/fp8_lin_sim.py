"""CPU simulation (torch) of the fp8 kernel's P encodings on the reference-generated fp8 goldens: the hardware e4m3 rounding of
P = 2^y against the log-linear byte b = sat_u8(rne(8 y + 56 - 8 delta)) (gen_fwd_x64_fp8.py "lin"). Dense attention, one pass,
m_ref = m_true - lag (lag = the lazy-rescale slack actually in use: 0 .. tau). Prints max / rms output errors against out_ref."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from helpers import FP8_CASES, GQA_FP8_CASES, load_dense_case, ref_tolerance

OFF, DELTA = 6.0, float(os.environ.get("DELTA", "0.0575"))
F8 = torch.float8_e4m3fn


def decode(b):
    b = b.to(torch.uint8)
    return b.view(F8).float()


def run(c, mode, lag):
    q, k, v = c["q"], c["k"], c["v"]
    B, Sq, H, D = q.shape
    Hk = k.shape[2]
    qd = c.get("q_descale", torch.ones(B, Hk)); kd = c.get("k_descale", torch.ones(B, Hk)); vd = c.get("v_descale", torch.ones(B, Hk))
    rep = H // Hk
    kk, vv = k.repeat_interleave(rep, 2), v.repeat_interleave(rep, 2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, kk)
    cs = (D ** -0.5) * math.log2(math.e) * (qd * kd).repeat_interleave(rep, 1)[:, :, None, None]
    m = s.max(-1, keepdim=True).values
    lagt = lag * torch.rand(m.shape, generator=torch.Generator().manual_seed(5)) if lag else 0.0
    y = (s - m) * cs + OFF + lagt           # m_ref = m_true - lagt / c
    if mode == "e4m3":
        p = torch.exp2(y).to(F8).float()
    elif mode == "lin":
        b = torch.clamp(torch.round(8 * y + 56 - 8 * DELTA), 0, 255)
        p = decode(b)
    else:
        p = torch.exp2(y)
    l = p.sum(-1, keepdim=True)
    o = torch.einsum("bhqk,bkhd->bqhd", p / l, vv) * vd.repeat_interleave(rep, 1)[:, None, :, None]
    lse = (m.squeeze(-1) * cs.squeeze(-1) + lagt.squeeze(-1) * 0 - 0) if False else None
    lse = (torch.log(l.squeeze(-1)) + ((m.squeeze(-1)) * cs.squeeze(-1) - (lagt.squeeze(-1) if lag else 0.0) - OFF) * math.log(2))
    return o, lse


for name in FP8_CASES + GQA_FP8_CASES:
    c = load_dense_case(name)
    tol = ref_tolerance(c["out_ref"], c["pt_maxerr"])
    print(f"{name}: reference rule {tol:.4f}, max|out_ref| {c['out_ref'].abs().max():.3f}")
    for lag in (0.0, 2.0):
        oe, le = run(c, "exact", lag)
        for mode in ("e4m3", "lin"):
            o, lse = run(c, mode, lag)
            e = (o - c["out_ref"]).abs()
            ee = (o - oe).abs()
            print(f"   lag {lag} {mode:5s}: vs out_ref max {e.max():.4f} rms {e.pow(2).mean().sqrt():.5f} | vs exact-P max {ee.max():.4f} rms {ee.pow(2).mean().sqrt():.5f}"
                  f" | lse err max {(lse - c['lse_ref']).abs().max():.4f} mean {(lse - c['lse_ref']).mean():+.5f}")
