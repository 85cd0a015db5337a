"""GPU: the head_dim-128 path on the OTHER matrix shape of gfx950 - v_mfma_f32_16x16x32 (liteattention_amd/csrc/gen_fwd_x64_m16.py,
round 5, A/B library build_variants/m16.so built on request: python -m liteattention_amd.build --m16). Not the product default (profiles/r05_m16.md: at parity,
+0.6 / -0.8 / -1.6 %), but a second, independently derived implementation of the same path - another register map, another cross-lane
scheme, another V image in LDS - that must pass the SAME parity tests against the same oracle: dense goldens, ragged grids, multi-step
lists bit-exact, fragmented lists, the headline-shape checks, fp16. One library per process, hence the subprocess.
Reference counterpart of the path: mainloop_fwd_sm90_tma_gmma_ws.hpp:1667-1755 + softmax.h:139-222."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M16 = os.path.join(ROOT, "build_variants", "m16.so")


def _run(args, timeout=900):
    env = dict(os.environ, LITEATTENTION_AMD_LIB=M16)
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


@pytest.fixture(scope="module", autouse=True)
def _library():
    if not os.path.exists(M16):     # an A/B library, not the product: its absence must not stop a `pytest -x` run of the product's tests
        pytest.skip("build_variants/m16.so is missing: python -m liteattention_amd.build --m16")
    code = ("import os, ctypes, torch; lib = ctypes.CDLL(%r); lib.la_build_info.restype = ctypes.c_char_p; print(lib.la_build_info().decode())" % M16)
    info = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
    assert "variant=1" in info and "wrong_results=0" in info and "m16" in info, info
    sys.path.insert(0, ROOT)
    from liteattention_amd import _buildinfo
    if _buildinfo.parse(info.strip()).get("src") != _buildinfo.source_hash():
        pytest.skip("build_variants/m16.so was built from other sources than this tree (stale): python -m liteattention_amd.build --m16")


def test_parity_suite_on_the_16x16x32_body():
    r = _run(["tests/test_gpu_parity.py", "tests/test_gpu_headline.py"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_lists_fp16_and_fragmented_lists_on_the_16x16x32_body():
    r = _run(["tests/test_gpu_fp16.py", "tests/test_gpu_fragmented.py", "-k", "not fp8 and not d64 and not d256 and not 64 and not 256"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_the_process_really_ran_the_16x16x32_kernel():
    """The kernel symbol is the same (la_fwd_x64_kernel<.., 128>): what tells the bodies apart is the instruction stream. The library on
    disk must hold 16x16x32 MFMAs in its gfx950 code object, and the product library none (it is all 32x32x16 / fp8 scaled)."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    import glob
    import shutil
    import tempfile
    counts = {}
    for name, lib in (("m16", M16), ("product", os.path.join(ROOT, "liteattention_amd", "libliteattention_amd.so"))):
        with tempfile.TemporaryDirectory() as d:
            copy = os.path.join(d, "lib.so")
            shutil.copy(lib, copy)
            subprocess.run([objdump, "--offloading", copy], capture_output=True, text=True, cwd=d)      # writes lib.so.<n>.hipv4-...-gfx950 beside the copy
            objs = glob.glob(os.path.join(d, "lib.so.*gfx950"))
            if not objs:
                pytest.skip("llvm-objdump --offloading extracted no gfx950 code object")
            dis = "".join(subprocess.run([objdump, "-d", o], capture_output=True, text=True).stdout for o in objs)
            counts[name] = (dis.count("v_mfma_f32_16x16x32_bf16"), dis.count("v_mfma_f32_32x32x16_bf16"))
    assert counts["m16"][0] >= 2 * 5 * 64 and counts["product"][0] == 0, counts       # two instantiations (SKIPABLE true / false) x 5 phases x 64
    assert counts["product"][1] > 0
