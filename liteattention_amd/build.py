"""In-tree build of the HIP extension: ``hipcc --offload-arch=gfx950 -shared`` -> libliteattention_amd.so.

Replaces the reference's nvcc/CUTLASS build (/root/reference/hopper/setup.py:381-674): no network, no
downloaded toolchain, no feature-flag matrix — the build is {bf16 / fp16: head_dim 64/96/128/192/256, fp8 e4m3: head_dim 128} for
gfx950 only (head dims in between run zero-padded on the next size up).
The .so is written next to this file so that it travels with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_NAME = "libliteattention_amd.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
SOURCES = ["la_fwd_kernel_v2.hip", "la_fwd_kernel_x64.hip", "la_prep_fp8.hip", "la_fwd_kernel_x64_fp8.hip", "la_aux_kernels.hip",
           "la_api.hip"]
HEADERS = ["la_kernel_params.h", "la_tiles.h", "la_fwd_common.h", "gen_fwd_x64.py", "gen_fwd_x64_fp8.py", "gen_epilogue.py"]
X64_GEN, X64_INC = "gen_fwd_x64.py", "la_fwd_x64_body.inc"      # the hand-scheduled main loop, included by la_fwd_kernel_x64.hip
X64_F16_INC = "la_fwd_x64_f16_body.inc"                        # the same generator with LA_X64_DTYPE=f16 (fp16 MFMA / conversions)
X64_BODIES = [(128, "bf16", X64_INC), (128, "f16", X64_F16_INC)] + [
    (d, t, f"la_fwd_x64_d{d}_{'f16_' if t == 'f16' else ''}body.inc") for d in (64, 96, 192, 256) for t in ("bf16", "f16")]   # LA_X64_D / LA_X64_DTYPE
X64F8_GEN, X64F8_INC = "gen_fwd_x64_fp8.py", "la_fwd_x64_fp8_body.inc"     # fp8: the same structure on the block-scaled MFMA
X64F8_EXP_INC = "la_fwd_x64_fp8_exp_body.inc"                               # LA_X64F8_OPT=exp: P = v_exp_f32 rounded by the hardware convert (LA_FLAG_EXACT_EXP)
X64F8_LVALU_INC = "la_fwd_x64_fp8_lvalu_body.inc"                           # LA_X64F8_OPT=lvalu: that, and fp32 row sums on the VALU (LA_FLAG_EXACT_ROWSUM)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(INCLUDE, "lite_attention_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, defines=(), out: str = None) -> str:
    """Compile every HIP source for gfx950 into one shared library. Returns its path.
    ``defines``/``out`` build an A/B variant (e.g. defines=["LA_NO_SETPRIO"], out="/path/lib_b.so") that
    LITEATTENTION_AMD_LIB can select; the default build has neither."""
    if out is not None:
        return _compile(out, defines, verbose)
    if not force and not is_stale():
        return LIB_PATH
    return _compile(LIB_PATH, (), verbose)


def _compile(lib_path: str, defines, verbose: bool) -> str:
    quiet = None if verbose else subprocess.DEVNULL
    for head_dim, dtype, inc in X64_BODIES:       # one generated body per (head dim, 16-bit element type)
        env = dict(os.environ, LA_X64_D=str(head_dim), LA_X64_DTYPE=dtype)
        if head_dim != 128:                        # LA_X64_OPT tunes the head_dim-128 body (tools/asm_variants.py); the others have their own knob
            env["LA_X64_OPT"] = os.environ.get(f"LA_X64_D{head_dim}_OPT", "")
            if head_dim == 64 and any(d.replace(" ", "") == "LA_D64_W2=1" for d in defines):
                env["LA_X64_OPT"] = os.environ.get("LA_X64_D64_OPT", "w2")       # -DLA_D64_W2=1 (A/B build): the two-waves-per-SIMD body
        subprocess.run([sys.executable, os.path.join(CSRC, X64_GEN), os.path.join(CSRC, inc)], check=True, stdout=quiet, env=env)
    env_f8 = dict(os.environ, LA_X64F8_OPT=os.environ.get("LA_X64F8_DEFAULT_OPT", ""))      # a global LA_X64F8_OPT must not leak into the default body
    subprocess.run([sys.executable, os.path.join(CSRC, X64F8_GEN), os.path.join(CSRC, X64F8_INC)], check=True, stdout=quiet, env=env_f8)
    for variant, inc in (("exp", X64F8_EXP_INC), ("lvalu", X64F8_LVALU_INC)):       # LA_X64F8_<VARIANT>_OPT tunes that body alone
        f8_opt = ",".join(x for x in (os.environ.get(f"LA_X64F8_{variant.upper()}_OPT", ""), variant) if x)
        subprocess.run([sys.executable, os.path.join(CSRC, X64F8_GEN), os.path.join(CSRC, inc)], check=True, stdout=quiet,
                       env=dict(os.environ, LA_X64F8_OPT=f8_opt))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I", INCLUDE, "-I", CSRC]
    cmd += [f"-D{d}" for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    tmp = lib_path + ".tmp"
    cmd += ["-o", tmp, "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, check=False, stderr=subprocess.PIPE, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)                       # the compiler's diagnostics, not just "non-zero exit status"
        raise RuntimeError(f"hipcc failed with exit status {res.returncode} (diagnostics above)")
    if verbose:
        sys.stderr.write("".join(ln + "\n" for ln in res.stderr.splitlines() if "remark:" not in ln and ln.strip()))
    _check_no_scratch(res.stderr, defines)
    os.replace(tmp, lib_path)
    return lib_path


def _check_no_scratch(remarks: str, defines) -> None:
    """The x64 kernels (a C++ shell around an asm body that clobbers nearly the whole register file) must not use scratch:
    hipcc (ROCm 7.2) has been seen to place the spill store of a value that is live across the body inside a finished
    divergent loop, where EXEC is 0 - the value is silently lost (wrong parameter words, wild addresses, GPU memory faults).
    The shells are written so that nothing per-lane is live across the body; this check keeps it that way."""
    name, bad, seen = None, [], 0
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize [bytes/lane]:" in line and name and "la_fwd_x64_" in name:
            size = int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])
            seen += 1
            if size != 0:
                bad.append((name, size))
    if seen == 0:      # a toolchain that words the remarks differently must not pass the gate silently
        raise RuntimeError("no kernel-resource-usage remark of an x64 kernel was parsed: the no-scratch check cannot run "
                           "(-Rpass-analysis=kernel-resource-usage output changed?)")
    if bad and not any(d.split("=")[0] == "LA_PROFILE_PHASES" for d in defines):
        raise RuntimeError("x64 kernels must not spill to scratch (see liteattention_amd/csrc/la_fwd_kernel_x64.hip, "
                           f"COMPILER HAZARD): {bad}")


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    print(build(force="--force" in sys.argv, verbose=True, defines=defs, out=outs[0] if outs else None))
