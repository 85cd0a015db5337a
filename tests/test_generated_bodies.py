"""CPU: static checks on the generated gfx950 bodies (liteattention_amd/csrc/gen_fwd_x64*.py). The bodies run inside a C++ shell
whose inline-asm statement declares what they clobber (v0-v222, s35-s95, every AGPR, m0, vcc, scc): a register outside that set
written by a body would silently corrupt compiler-owned state. Also pins the MFMA counts per step and that the head_dim-128 bodies of
the two 16-bit types differ in nothing but the MFMA / convert opcodes."""
import os
import re
import subprocess
import sys

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "liteattention_amd", "csrc")
CASES = [("gen_fwd_x64.py", {"LA_X64_D": str(d), "LA_X64_DTYPE": t}) for d in (96, 128, 192, 256) for t in ("bf16", "f16")] + \
        [("gen_fwd_x64.py", {"LA_X64_D": str(d), "LA_X64_DTYPE": "bf16", "LA_X64_FORM": "half"}) for d in (64, 96, 128)] + \
        [("gen_fwd_x64_fp8.py", dict({"LA_X64F8_D": str(d)}, **({"LA_X64F8_OPT": o} if o else {}))) for d in (64, 96, 128, 192, 256) for o in ("", "exp", "lvalu")]


def _generate(tmp_path, gen, env):
    out = tmp_path / "body.inc"
    e = dict(os.environ, **env)
    e.pop("LA_X64_OPT", None)
    if "LA_X64F8_OPT" not in env:
        e.pop("LA_X64F8_OPT", None)
    if gen == "gen_fwd_x64_fp8.py":         # the generator checks the body's NAME against its head dim and form of P
        d, o = env.get("LA_X64F8_D", "128"), env.get("LA_X64F8_OPT", "")
        out = tmp_path / ("la_fwd_x64_fp8_" + ("" if d == "128" else f"d{d}_") + (o + "_" if o else "") + "body.inc")
    subprocess.run([sys.executable, os.path.join(CSRC, gen), str(out)], check=True, stdout=subprocess.DEVNULL, env=e)
    return out.read_text()


def _registers(text):
    """(max VGPR, max AGPR, set of SGPRs) named anywhere in the body."""
    vmax = amax = -1
    sgprs = set()
    for kind, lo, hi, single in re.findall(r"\b([vas])(?:\[(\d+):(\d+)\]|(\d+))\b", text):
        first, last = (int(single), int(single)) if single else (int(lo), int(hi))
        if kind == "v":
            vmax = max(vmax, last)
        elif kind == "a":
            amax = max(amax, last)
        else:
            sgprs.update(range(first, last + 1))
    return vmax, amax, sgprs


@pytest.mark.parametrize("gen,env", CASES, ids=[f"{g[:-3]}-{'-'.join(e.values())}" for g, e in CASES])
def test_body_stays_inside_the_declared_clobbers(tmp_path, gen, env):
    text = _generate(tmp_path, gen, env)
    body = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith((";", "//")))
    vmax, amax, sgprs = _registers(body)
    assert 0 <= vmax <= 222, vmax                      # LA_X64_CLOBBERS: v0 .. v222
    assert amax <= 255
    assert sgprs and min(sgprs) >= 35 and max(sgprs) <= 95, (min(sgprs), max(sgprs))      # s32-s34 are ABI-reserved, s0-s31 the shell's
    assert "%0" in body and "%1" in body               # the two inputs: wave index, LDS address of the parameter block
    assert "s_setpc" not in body and "s_endpgm" not in body
    # every label is local to the asm statement (%= suffix): two instantiations in one translation unit must not collide
    for lab in re.findall(r"^\s*([.\w%=]+):\s*$", body, flags=re.M):
        assert lab.endswith("%="), lab


@pytest.mark.parametrize("D,per_phase", [(96, 24), (128, 32), (192, 24), (256, 32)])
def test_mfma_count_per_step(tmp_path, D, per_phase):
    """prologue QK of tile 0 (one phase) + two unrolled steps of (QK + PV): 5 phases of MFMAs."""
    text = _generate(tmp_path, "gen_fwd_x64.py", {"LA_X64_D": str(D)})
    assert text.count("v_mfma_f32_32x32x16_bf16") == 5 * per_phase
    assert "v_mfma_f32_32x32x16_f16" not in text


@pytest.mark.parametrize("D,qk,pv", [(64, 4, 4), (96, 8, 6), (128, 8, 8), (192, 6, 6), (256, 8, 8)])
@pytest.mark.parametrize("form", ["", "exp", "lvalu"])
def test_fp8_mfma_count_per_step(tmp_path, D, qk, pv, form):
    """fp8 bodies (gen_fwd_x64_fp8.py, LA_X64F8_D): prologue QK of tile 0 + two unrolled steps of (QK + PV [+ one row-sum MFMA per q-block in the
    matrix-pipe row-sum forms]); 64-row bodies (64 / 128) hold two q-blocks per wave, the 192 / 256 bodies one. The head_dim-192 body clamps the
    K tile's DMA source chunks to the 12 that exist."""
    text = _generate(tmp_path, "gen_fwd_x64_fp8.py", dict({"LA_X64F8_D": str(D)}, **({"LA_X64F8_OPT": form} if form else {})))
    rowsum = 0 if form == "lvalu" else (2 if D <= 128 else 1)
    assert text.count("v_mfma_scale_f32_32x32x64_f8f6f4") == qk + 2 * (qk + pv + rowsum)
    assert ("v_min_u32" in text) == (D in (96, 192))
    assert text.count("s_barrier") == 2 + 2


def test_fp16_body_differs_only_in_the_type_dependent_opcodes(tmp_path):
    a = _generate(tmp_path, "gen_fwd_x64.py", {"LA_X64_D": "128", "LA_X64_DTYPE": "bf16"})
    b = _generate(tmp_path, "gen_fwd_x64.py", {"LA_X64_D": "128", "LA_X64_DTYPE": "f16"})
    norm = lambda t: t.replace("v_mfma_f32_32x32x16_bf16", "MFMA").replace("v_mfma_f32_32x32x16_f16", "MFMA") \
                      .replace("v_cvt_pk_bf16_f32", "CVT").replace("v_cvt_pk_f16_f32", "CVT")                      # noqa: E731
    la, lb = norm(a).splitlines(), norm(b).splitlines()
    assert len(la) == len(lb)
    assert [x for x, y in zip(la, lb) if x != y and not x.lstrip().startswith(("//", ";"))] == []


def test_two_waves_per_simd_body_of_head_dim_64_fits_two_waves(tmp_path):
    """The A/B body of round 4 (gen_fwd_x64.py LA_X64_OPT=w2, -DLA_D64_W2=1; profiles/r04_head_dim_64.md): an 8-wave workgroup with two
    waves per SIMD needs <= 256 registers per wave INCLUDING what the C++ shell keeps across the body: the body stays inside v0-v89 +
    a0-a79 (LA_X64W2_CLOBBERS), has 8 + 8 MFMAs per step and two loops (waves 0-3 / waves 4-7, the latter one QK ahead)."""
    out = tmp_path / "w2.inc"
    e = dict(os.environ, LA_X64_D="64", LA_X64_OPT="w2")
    subprocess.run([sys.executable, os.path.join(CSRC, "gen_fwd_x64.py"), str(out)], check=True, stdout=subprocess.DEVNULL, env=e)
    text = out.read_text()
    body = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith((";", "//")))
    vmax, amax, sgprs = _registers(body)
    assert 0 <= vmax <= 89 and 0 <= amax <= 79, (vmax, amax)
    assert min(sgprs) >= 35 and max(sgprs) <= 95
    # group a: 2 steps x 16; group b: QK of tile 0 (8) + 2 steps x 16
    assert body.count("v_mfma_f32_32x32x16_bf16") == 2 * 16 + 8 + 2 * 16
    assert body.count("s_barrier") == 1 + 2 + 2          # prologue + one per step copy, both groups: every wave meets the same barriers
    shell = open(os.path.join(CSRC, "la_fwd_kernel_x64.hip")).read()
    clob = shell.split("#define LA_X64W2_CLOBBERS")[1].split("namespace la")[0]
    assert '"v89"' in clob and '"v90"' not in clob and '"a79"' in clob and '"a80"' not in clob


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_m16_body_stays_inside_the_declared_clobbers_and_has_the_16x16x32_counts(tmp_path, dtype):
    """The head_dim-128 body on v_mfma_f32_16x16x32 (gen_fwd_x64_m16.py, round 5; A/B build -DLA_X64_M16=1): same shell, same clobber
    set; 64 + 64 MFMAs per step (prologue QK + two unrolled steps = 5 phases of 64), no 32x32x16 MFMA, 64-bit VGPR tuples even-aligned."""
    out = tmp_path / "m16.inc"
    e = dict(os.environ, LA_X64_DTYPE=dtype)
    e.pop("LA_X64_OPT", None)
    subprocess.run([sys.executable, os.path.join(CSRC, "gen_fwd_x64_m16.py"), str(out)], check=True, stdout=subprocess.DEVNULL, env=e)
    text = out.read_text()
    assert "la_body_options: m16; wrong_results=0" in text.splitlines()[1]
    body = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith((";", "//")))
    vmax, amax, sgprs = _registers(body)
    assert 0 <= vmax <= 222 and amax <= 255 and min(sgprs) >= 35 and max(sgprs) <= 95
    assert max(int(hi) for _, hi in re.findall(r"\ba\[(\d+):(\d+)\]", body)) == 255 and max(int(hi) for _, hi in re.findall(r"\bv\[(\d+):(\d+)\]", body)) <= 222
    assert body.count(f"v_mfma_f32_16x16x32_{dtype}") == 5 * 64 and "32x32x16" not in body
    assert body.count("v_permlane16_swap_b32") >= 2 * 2 + 1                     # the transposing row reduction, twice per unrolled loop + prologue
    for lo in re.findall(r"\bv\[(\d+):(\d+)\]", body):
        assert int(lo[0]) % 2 == 0, lo                                             # gfx950: VGPR tuples must be 64-bit aligned
    for lab in re.findall(r"^\s*([.\w%=]+):\s*$", body, flags=re.M):
        assert lab.endswith("%="), lab
