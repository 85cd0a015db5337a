"""CPU: the product build cannot be bent by the environment.

The body generators honour pricing options that produce WRONG results on purpose (``nosoftmax``, ``mfmasum``, ``mfma16:*``,
``halfskip:*``, ``vfake`` ...; tools/asm_variants.py). Round 4's build forwarded a global ``LA_X64_OPT`` into the default head_dim-128
body. Now: the product build strips every ``LA_X64*`` variable from the generators' environment, every body carries a tag line saying
which options went in and whether they change results, the library exports ``la_build_info()``, and ``_cabi.load()`` refuses a
default-location library whose record says variant / wrong results / other sources. The reference has no counterpart (its feature
flags live in setup.py's environment, hopper/setup.py:47-68)."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "liteattention_amd")
POISON = {"LA_X64_OPT": "nosoftmax,mfmasum", "LA_X64_D64_OPT": "w2", "LA_X64_D256_OPT": "nobarrier", "LA_X64F8_OPT": "nomx",
          "LA_X64F8_DEFAULT_OPT": "nobarrier", "LA_X64F8_EXP_OPT": "halfbarrier", "LA_X64F8_LVALU_OPT": "nowaitvm", "LA_X64_FORM": "half", "LA_X64_HALF_OPT": "nosoftmax", "LA_X64F8_D": "64",
          "LA_X64F8_D64_LVALU_OPT": "nobarrier"}


def _generate(tmp, variant, env):
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "import importlib.util as u\n"
            "s = u.spec_from_file_location('la_build_t', %r); b = u.module_from_spec(s); s.loader.exec_module(b)\n"
            "g, m = b.generate_bodies(%r, %r)\n"
            "try:\n    rec = b._build_record(g, (), %r)\nexcept RuntimeError as e:\n    rec = 'ERROR ' + str(e)\n"
            "print(json.dumps({'generated': g, 'macros': m, 'record': rec}))\n") % (ROOT, os.path.join(PKG, "build.py"), str(tmp), variant, variant)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    return json.loads(res.stdout.splitlines()[-1])


def test_a_poisoned_environment_yields_the_clean_product_bodies(tmp_path):
    clean_dir, dirty_dir = tmp_path / "clean", tmp_path / "dirty"
    clean_dir.mkdir(); dirty_dir.mkdir()
    a = _generate(clean_dir, False, {})
    b = _generate(dirty_dir, False, POISON)
    assert len(a["generated"]) == 31 and [os.path.basename(p) for p in a["generated"]] == [os.path.basename(p) for p in b["generated"]]
    for pa, pb in zip(a["generated"], b["generated"]):
        assert open(pa, "rb").read() == open(pb, "rb").read(), os.path.basename(pa)        # byte for byte
        head = open(pb).read(400)
        assert "wrong_results=0" in head and "// la_body_options:" in head
    for r in (a["record"], b["record"]):
        assert ";variant=0;wrong_results=0;opts=" in r and r.endswith("opts=") and not r.startswith("ERROR")
    assert a["macros"] == [] and b["macros"] == []                                      # the product shells include the tree's own bodies


def test_a_variant_honours_the_options_and_says_so(tmp_path):
    v = _generate(tmp_path, True, {"LA_X64_OPT": "nosoftmax", "LA_X64_D64_OPT": "x:3"})
    d128 = [p for p in v["generated"] if os.path.basename(p) == "la_fwd_x64_body.inc"][0]
    assert "wrong_results=1" in open(d128).read(400) and "nosoftmax" in open(d128).read(400)
    d64 = [p for p in v["generated"] if os.path.basename(p) == "la_fwd_x64_d64_body.inc"][0]
    assert "la_body_options: x:3; wrong_results=0" in open(d64).read(400)               # a schedule-only option: results unchanged
    assert ";variant=1;wrong_results=1;" in v["record"] and "nosoftmax" in v["record"] and "x:3" in v["record"]
    assert any(m.startswith("-DLA_X64_BODY_INC=") and str(tmp_path) in m for m in v["macros"])
    assert any(m.startswith("-DLA_X64F8_LVALU_CONSTS_INC=") for m in v["macros"])


def test_the_product_build_refuses_defines_and_dirty_bodies():
    import importlib.util as u
    s = u.spec_from_file_location("la_build_t2", os.path.join(PKG, "build.py"))
    b = u.module_from_spec(s)
    s.loader.exec_module(b)
    with pytest.raises(ValueError, match="no -D defines"):
        b.build(defines=["LA_SCHED_GANG=1"])
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "la_fwd_x64_body.inc")
        open(p, "w").write("// GENERATED\n// la_body_options: nosoftmax; wrong_results=1 (PRICING ONLY, results are wrong: nosoftmax)\n")
        with pytest.raises(RuntimeError, match="option-free"):
            b._build_record([p], (), False)
        assert ";variant=1;wrong_results=1;opts=d128[nosoftmax]" in b._build_record([p], (), True)


def test_the_loaded_library_is_the_product_build_of_this_tree():
    from liteattention_amd import _buildinfo, _cabi
    info = _cabi.build_info()
    assert info["abi"] == str(_cabi.LA_ABI_VERSION) and info["variant"] == "0" and info["wrong_results"] == "0" and info["opts"] == ""
    assert info["src"] == _buildinfo.source_hash()
    rec = _buildinfo.record_in_file(_cabi.LIB_PATH)                      # the same record, read without loading (build.is_stale)
    assert rec == {k: info[k] for k in ("src", "variant", "wrong_results", "opts")}


@pytest.mark.parametrize("record,match", [
    ("abi=7;src=%s;variant=1;wrong_results=0;opts=-DLA_SCHED_GANG=1", "variant"),
    ("abi=7;src=%s;variant=1;wrong_results=1;opts=d128[nosoftmax]", "variant"),
    ("abi=7;src=0123456789abcdef;variant=0;wrong_results=0;opts=", "other sources"),
])
def test_load_refuses_a_default_location_library_with_the_wrong_record(monkeypatch, record, match):
    from liteattention_amd import _buildinfo, _cabi
    if "%s" in record:
        record = record % _buildinfo.source_hash()
    fake = types.SimpleNamespace(la_build_info=lambda: record.encode())
    monkeypatch.delenv("LITEATTENTION_AMD_LIB", raising=False)
    with pytest.raises(_cabi.NativeLibraryError, match=match):
        _cabi._check_build_record(fake)
    monkeypatch.setenv("LITEATTENTION_AMD_LIB", "/some/variant.so")      # naming a file is the one way to run a variant
    _cabi._check_build_record(fake)


def test_every_loop_head_sits_at_its_pinned_code_placement():
    """Round 5 (profiles/r05_code_placement.md): where the loop head of a generated body falls inside a 32-byte fetch window moves the
    kernel by up to 2-3 % with a period of 32 bytes, and until now that phase was whatever the C++ shell in front of the asm statement
    left behind (an edit to the list writer moved the headline kernel from phase 8 to 16: -0.8 %). The generators pin it (.p2align 5 +
    PHASE / 4 s_nop); this test reads the phases back from the built library's gfx950 code objects."""
    import glob
    import re
    import shutil
    import tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(PKG, "libliteattention_amd.so")
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("llvm-objdump or the library is not there")
    # (the fp8 names carry their head dim too - <lists, form of P, head dim> - so their keys are tried first; a form's phase is the same at 64 and 128)
    want = {"fp8_kernelILb1ELi0E": 0, "fp8_kernelILb0ELi0E": 0, "fp8_kernelILb1ELi1E": 24, "fp8_kernelILb0ELi1E": 24, "fp8_kernelILb1ELi2E": 8,
            "fp8_kernelILb0ELi2E": 8, "Li64E": 24, "Li96E": 8, "Li128E": 8, "Li192E": 8, "Li256E": 0}
    seen = 0
    with tempfile.TemporaryDirectory() as d:
        copy = os.path.join(d, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([objdump, "--offloading", copy], capture_output=True, text=True, cwd=d)
        for obj in glob.glob(os.path.join(d, "lib.so.*gfx950")):
            dis = subprocess.run([objdump, "-d", obj], capture_output=True, text=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <(_ZN2la[^>]*la_fwd_x64[^>]*)>:", dis, re.M):
                seg = dis[m.end():]
                nxt = re.search(r"^[0-9a-f]+ <", seg, re.M)
                seg = seg[:nxt.start()] if nxt else seg
                heads = re.findall(r"s_cmp_lt_u32 s63, s55\s+// ([0-9A-Fa-f]+):", seg)          # the loop test: S_I < S_NTILES
                if "fp8" not in m.group(1) and re.search(r"Li\d+ELb1E", m.group(1)):          # the half-vote form: its loop head is the test of the step form (the loop test sits in front of the drain)
                    heads = re.findall(r"s_cmp_eq_u32 s81, 3\s+// ([0-9A-Fa-f]+):", seg)
                if not heads:
                    continue
                key = next(k for k in want if k in m.group(1))
                assert int(heads[0], 16) % 32 == want[key], (m.group(1), int(heads[0], 16) % 32, want[key])
                seen += 1
    assert seen == 56, seen            # 5 head dims x 2 element types x 2 (lists / dense) + the half-vote form of head dims 64 / 96 / 128 x 2 element types + 3 fp8 forms x 2 x 5 head dims
