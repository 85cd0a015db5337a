"""In-tree build of the HIP extension: ``hipcc --offload-arch=gfx950 -shared`` -> libliteattention_amd.so.

Replaces the reference's nvcc/CUTLASS build (/root/reference/hopper/setup.py:381-674): no network, no
downloaded toolchain, no feature-flag matrix — the build is {bf16 / fp16: head_dim 64/96/128/192/256, fp8 e4m3: head_dim 64/96/128/192/256} for
gfx950 only (head dims in between run zero-padded on the next size up).
The .so is written next to this file so that it travels with the source tree.
"""
from __future__ import annotations

import contextlib
import fcntl
import os
import shutil
import subprocess
import sys
import tempfile

try:
    from . import _buildinfo
except ImportError:                       # loaded by path (__graft_entry__.build(), before the package can be imported)
    import importlib.util as _ilu
    _spec = _ilu.spec_from_file_location("la_buildinfo", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_buildinfo.py"))
    _buildinfo = _ilu.module_from_spec(_spec)
    _spec.loader.exec_module(_buildinfo)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_NAME = "libliteattention_amd.so"
LIB_PATH = os.path.join(PKG_DIR, LIB_NAME)
SOURCES, HEADERS = _buildinfo.SOURCES, _buildinfo.HEADERS
X64_M16_GEN = "gen_fwd_x64_m16.py"                              # head_dim 128 on the 16x16x32 MFMA (A/B build -DLA_X64_M16=1)
X64_GEN, X64_INC = "gen_fwd_x64.py", "la_fwd_x64_body.inc"      # the hand-scheduled main loop, included by la_fwd_kernel_x64.hip
X64_F16_INC = "la_fwd_x64_f16_body.inc"                        # the same generator with LA_X64_DTYPE=f16 (fp16 MFMA / conversions)
X64_BODIES = [(128, "bf16", X64_INC), (128, "f16", X64_F16_INC)] + [
    (d, t, f"la_fwd_x64_d{d}_{'f16_' if t == 'f16' else ''}body.inc") for d in (64, 96, 192, 256) for t in ("bf16", "f16")]   # LA_X64_D / LA_X64_DTYPE
X64_HALF_INC, X64_HALF_F16_INC = "la_fwd_x64_half_body.inc", "la_fwd_x64_half_f16_body.inc"      # LA_X64_FORM=half: the half-vote form of head_dim 128
X64F8_GEN, X64F8_INC = "gen_fwd_x64_fp8.py", "la_fwd_x64_fp8_body.inc"     # fp8: the same structure on the block-scaled MFMA
X64F8_EXP_INC = "la_fwd_x64_fp8_exp_body.inc"                               # LA_X64F8_OPT=exp: P = v_exp_f32 rounded by the hardware convert (LA_FLAG_FP8_MFMA_ROWSUM)
X64F8_LVALU_INC = "la_fwd_x64_fp8_lvalu_body.inc"                           # LA_X64F8_OPT=lvalu: that, and fp32 row sums on the VALU (the DEFAULT fp8 form: the reference's arithmetic)


def body_macro(head_dim: int, dtype: str) -> str:
    """Name of the shell's include macro for a bf16 / fp16 body (la_fwd_kernel_x64.hip)."""
    return "LA_X64_" + ("" if head_dim == 128 else f"D{head_dim}_") + ("F16_" if dtype == "f16" else "") + "BODY_INC"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    """The product library is missing, older than a build input, or its own record (la_build_info) names other sources / a variant."""
    if not os.path.exists(LIB_PATH):
        return True
    rec = _buildinfo.record_in_file(LIB_PATH)
    if rec is None or rec["variant"] != "0" or rec["wrong_results"] != "0" or rec["src"] != _buildinfo.source_hash():
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(INCLUDE, "lite_attention_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False, defines=(), out: str = None) -> str:
    """Compile every HIP source for gfx950 into one shared library. Returns its path.

    The PRODUCT library (``out`` is None -> liteattention_amd/libliteattention_amd.so) is built from the default bodies with no
    define, whatever the environment holds: every ``LA_X64*`` generator option variable is removed from the generators' environment
    and ``defines`` is refused (a stray LA_X64_OPT=nosoftmax at build time used to yield a default library that
    computes garbage). ``defines`` / generator options only go into an A/B or pricing VARIANT, ``out="/path/lib_b.so"``, whose bodies
    are generated beside it (``<out>.gen/``, the tree's own bodies stay the default ones), whose ``la_build_info()`` says ``variant=1``
    (+ ``wrong_results=1`` when an option that changes the arithmetic went in) and which only ``LITEATTENTION_AMD_LIB`` can select."""
    if out is not None:
        return _compile(out, defines, verbose, variant=True)
    if defines:
        raise ValueError("the product library takes no -D defines: build a variant with out=... (python -m liteattention_amd.build -D... --out=...)")
    if not force and not is_stale():
        return LIB_PATH
    with _build_lock(LIB_PATH):                    # ranks / xdist workers starting together: one builds, the others find it fresh
        if not force and not is_stale():
            return LIB_PATH
        return _compile(LIB_PATH, (), verbose, variant=False)


@contextlib.contextmanager
def _build_lock(lib_path: str):
    """Exclusive advisory lock per output library (``<lib>.lock``, git-ignored): concurrent builds of one target serialise."""
    fd = os.open(lib_path + ".lock", os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


M16_VARIANT = os.path.join(os.path.dirname(PKG_DIR), "build_variants", "m16.so")      # the head_dim-128 bodies on v_mfma_f32_16x16x32 (A/B build)


def build_m16_variant(force: bool = False, verbose: bool = False) -> str:
    """The 16x16x32-MFMA body of head_dim 128 as an A/B library beside the product one (``build_variants/m16.so``: git-ignored, travels
    to the GPU box like every built .so). ``__graft_entry__.build()`` builds it so that the round-end GPU tests can run the parity suite
    on it (tests/test_gpu_m16.py, in a subprocess with LITEATTENTION_AMD_LIB). Never the product: its record says variant=1."""
    rec = _buildinfo.record_in_file(M16_VARIANT) if os.path.exists(M16_VARIANT) else None
    fresh = rec is not None and rec["src"] == _buildinfo.source_hash() and rec["variant"] == "1" and "m16" in rec["opts"] and rec["wrong_results"] == "0"
    # its own generator is hashed into no record (an edit to the A/B body must not invalidate the PRODUCT library): compare times
    fresh = fresh and os.path.getmtime(os.path.join(CSRC, X64_M16_GEN)) <= os.path.getmtime(M16_VARIANT)
    if fresh and not force:
        return M16_VARIANT
    os.makedirs(os.path.dirname(M16_VARIANT), exist_ok=True)
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("LA_X64")}      # the variant of record: every body at its default schedule
    try:
        with _build_lock(M16_VARIANT):
            return _compile(M16_VARIANT, ["LA_X64_M16=1"], verbose, variant=True)
    finally:
        os.environ.update(saved)


def _generator_env(variant: bool) -> dict:
    """Environment of the body generators. Product build: no LA_X64* variable survives (LA_X64_OPT, LA_X64_D<D>_OPT, LA_X64F8_OPT,
    LA_X64F8_<FORM>_OPT, LA_X64F8_DEFAULT_OPT, LA_X64_D, LA_X64_DTYPE - the last two are set per body below)."""
    if variant:
        return dict(os.environ)
    return {k: v for k, v in os.environ.items() if not k.startswith("LA_X64")}


def generate_bodies(gen_dir: str, variant: bool, defines=(), quiet=subprocess.DEVNULL):
    """Run the body generators into ``gen_dir``. Returns (paths of the generated bodies, -D macros that point the shells at them).
    Product build (``variant`` False): the generators see no LA_X64* option, whatever this process's environment holds."""
    base_env = _generator_env(variant)
    generated, macros = [], []

    def generate(gen, inc, env, macro, consts_macro=None):
        path = os.path.join(gen_dir, inc)
        subprocess.run([sys.executable, os.path.join(CSRC, gen), path], check=True, stdout=quiet, env=env)
        generated.append(path)
        if variant:
            macros.append(f'-D{macro}="{path}"')
            if consts_macro:
                macros.append(f'-D{consts_macro}="{path.replace("_body.inc", "_consts.h")}"')

    for head_dim, dtype, inc in X64_BODIES:       # one generated body per (head dim, 16-bit element type)
        env = dict(base_env, LA_X64_D=str(head_dim), LA_X64_DTYPE=dtype)
        if variant and head_dim != 128:            # LA_X64_OPT tunes the head_dim-128 body (tools/asm_variants.py); the others have their own knob
            env["LA_X64_OPT"] = os.environ.get(f"LA_X64_D{head_dim}_OPT", "")
            if head_dim == 64 and any(d.replace(" ", "") == "LA_D64_W2=1" for d in defines):
                env["LA_X64_OPT"] = os.environ.get("LA_X64_D64_OPT", "w2")       # -DLA_D64_W2=1 (A/B build): the two-waves-per-SIMD body
        gen = X64_GEN
        if variant and head_dim == 128 and any(d.replace(" ", "") == "LA_X64_M16=1" for d in defines):
            gen = X64_M16_GEN                       # -DLA_X64_M16=1 (A/B build): head_dim 128 on v_mfma_f32_16x16x32 (LA_X64_OPT tunes it)
        generate(gen, inc, env, body_macro(head_dim, dtype))
    for head_dim in (128, 64, 96):                         # skip lists per 128-row half (LA_FLAG_HALF_VOTE): the 256-row kernels
        for dtype in ("bf16", "f16"):
            stem = ("" if head_dim == 128 else f"d{head_dim}_") + "half_" + ("f16_" if dtype == "f16" else "")
            env = dict(base_env, LA_X64_D=str(head_dim), LA_X64_DTYPE=dtype, LA_X64_FORM="half")
            if variant:
                env["LA_X64_OPT"] = os.environ.get("LA_X64_HALF_OPT" if head_dim == 128 else f"LA_X64_D{head_dim}_HALF_OPT", "")
            generate(X64_GEN, f"la_fwd_x64_{stem}body.inc", env, "LA_X64_" + stem.upper() + "BODY_INC")
    f8_default = os.environ.get("LA_X64F8_DEFAULT_OPT", "") if variant else ""      # a global LA_X64F8_OPT never reaches the default body
    generate(X64F8_GEN, X64F8_INC, dict(base_env, LA_X64F8_OPT=f8_default), "LA_X64F8_BODY_INC", "LA_X64F8_CONSTS_INC")
    for form, inc in (("exp", X64F8_EXP_INC), ("lvalu", X64F8_LVALU_INC)):       # LA_X64F8_<FORM>_OPT tunes that body alone (variants only)
        extra = os.environ.get(f"LA_X64F8_{form.upper()}_OPT", "") if variant else ""
        generate(X64F8_GEN, inc, dict(base_env, LA_X64F8_OPT=",".join(x for x in (extra, form) if x)),
                 f"LA_X64F8_{form.upper()}_BODY_INC", f"LA_X64F8_{form.upper()}_CONSTS_INC")
    for head_dim in (64, 96, 192, 256):           # the other head dims (round 6): the three forms of P again, LA_X64F8_D=<head dim>
        for form in ("", "exp", "lvalu"):
            stem = f"d{head_dim}_" + (form + "_" if form else "")
            extra = os.environ.get(f"LA_X64F8_D{head_dim}_{form.upper() or 'DEFAULT'}_OPT", "") if variant else ""
            generate(X64F8_GEN, f"la_fwd_x64_fp8_{stem}body.inc",
                     dict(base_env, LA_X64F8_D=str(head_dim), LA_X64F8_OPT=",".join(x for x in (extra, form) if x)), f"LA_X64F8_{stem.upper()}BODY_INC")
    return generated, macros


def _compile(lib_path: str, defines, verbose: bool, variant: bool = False) -> str:
    gen_dir = CSRC
    if variant:                                    # a variant's bodies live beside it; the tree keeps the product bodies
        gen_dir = os.path.abspath(lib_path) + ".gen"
        os.makedirs(gen_dir, exist_ok=True)
    generated, macros = generate_bodies(gen_dir, variant, defines, None if verbose else subprocess.DEVNULL)
    info = _build_record(generated, defines, variant)
    # one hipcc -c per source, in parallel (the sources share no device code: every kernel is launched from its own file), then one link
    common = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, f'-DLA_BUILD_INFO="{info}"']
    common += [f"-D{d}" for d in defines] + macros
    # objects and the link output of THIS build only: two builds of one target never share (or delete) each other's files
    out_dir = os.path.dirname(os.path.abspath(lib_path))
    obj_dir = tempfile.mkdtemp(prefix=os.path.basename(lib_path) + ".obj.", dir=out_dir)
    fd, tmp = tempfile.mkstemp(prefix=os.path.basename(lib_path) + ".tmp.", dir=out_dir)
    os.close(fd)
    try:
        return _compile_into(lib_path, obj_dir, tmp, common, defines, verbose)
    finally:
        shutil.rmtree(obj_dir, ignore_errors=True)
        if os.path.exists(tmp):
            os.unlink(tmp)


def _compile_into(lib_path: str, obj_dir: str, tmp: str, common, defines, verbose: bool) -> str:

    def compile_one(src):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = common + ["-c", os.path.join(CSRC, src), "-o", obj, "-Rpass-analysis=kernel-resource-usage"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        return obj, subprocess.run(cmd, check=False, stderr=subprocess.PIPE, text=True)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    remarks = ""
    for (obj, res), src in zip(results, SOURCES):
        if res.returncode != 0:
            sys.stderr.write(res.stderr)                   # the compiler's diagnostics, not just "non-zero exit status"
            raise RuntimeError(f"hipcc failed on {src} with exit status {res.returncode} (diagnostics above)")
        if verbose:
            sys.stderr.write("".join(ln + "\n" for ln in res.stderr.splitlines() if "remark:" not in ln and ln.strip()))
        remarks += res.stderr
    link = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for obj, _ in results] + ["-o", tmp],
                          check=False, stderr=subprocess.PIPE, text=True)
    if link.returncode != 0:
        sys.stderr.write(link.stderr)
        raise RuntimeError(f"hipcc failed to link with exit status {link.returncode} (diagnostics above)")
    _check_no_scratch(remarks, defines)
    os.chmod(tmp, 0o755)
    os.replace(tmp, lib_path)
    return lib_path


def _build_record(generated, defines, variant: bool) -> str:
    """The text behind la_build_info(): source hash, variant / wrong_results, and the options that went in - read back from the tag
    line every generator writes at the top of its body (``// la_body_options: ...; wrong_results=N``), so the record says what the
    bodies ARE, not what the environment asked for."""
    opts, wrong = [], False
    for path in generated:
        with open(path) as f:
            tag = [ln for ln in (f.readline(), f.readline()) if ln.startswith("// la_body_options:")]
        if not tag:
            raise RuntimeError(f"{path}: no la_body_options tag (generator out of date?)")
        body_opts = tag[0].split(":", 1)[1].split(";")[0].strip()
        wrong = wrong or "wrong_results=1" in tag[0]
        name = os.path.basename(path).replace("la_fwd_x64_", "").replace("_body.inc", "").replace("body.inc", "d128")
        if body_opts not in ("-", "exp", "lvalu") or not name:
            opts.append(f"{name}[{body_opts}]")
    opts += [f"-D{d}" for d in defines]
    if not variant and (opts or wrong):
        raise RuntimeError(f"the product build must be option-free, got {opts} (wrong_results={wrong}): generator environment not clean?")
    text = ",".join(opts).replace(";", ",").replace('"', "'").replace(" ", "")
    return f"src={_buildinfo.source_hash()};variant={1 if variant else 0};wrong_results={1 if wrong else 0};opts={text}"


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
    if "--m16" in sys.argv:                        # the closed round-5 A/B library (tests/test_gpu_m16.py runs the parity suite on it when it exists)
        print(build_m16_variant(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, defines=defs, out=outs[0] if outs else None))
