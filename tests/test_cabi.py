"""CPU: the C-ABI shared library loads, exports every symbol include/*.h declares, and its argument
struct matches the ctypes mirror. No compute is launched (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from liteattention_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lite_attention_amd.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int64_t|int|char\s*\*|const char\*)\s+\**\s*(la_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_expected_entry_points():
    assert declared_functions() == sorted(_cabi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _cabi.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.la_abi_version() == _cabi.LA_ABI_VERSION


def test_struct_layout_matches_header(tmp_path):
    """Compile a C program against the header with gcc: sizeof/offsetof must equal the ctypes mirror
    (also proves the header is plain C)."""
    fields = [f[0] for f in _cabi.LaFwdArgs._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){",
            'printf("%zu\\n", sizeof(la_fwd_args));']
    prog += [f'printf("%zu\\n", offsetof(la_fwd_args, {f}));' for f in fields]
    prog += ["return 0;}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(_cabi.LaFwdArgs)
    for f, off in zip(fields, out[1:]):
        assert getattr(_cabi.LaFwdArgs, f).offset == int(off), f


def test_tile_sizes_and_status_strings():
    assert _cabi.get_tile_sizes(128, 2) == (256, 64) and _cabi.get_tile_sizes(64, 2) == (256, 64)
    assert _cabi.get_tile_sizes(96, 2) == (256, 64) and _cabi.get_tile_sizes(192, 2) == (128, 64) and _cabi.get_tile_sizes(256, 2) == (128, 64)
    assert _cabi.get_tile_sizes(128, 1) == (256, 64)
    lib = _cabi.load()
    m, n = ctypes.c_int(), ctypes.c_int()
    assert lib.la_get_tile_sizes(48, 2, ctypes.byref(m), ctypes.byref(n)) == _cabi.LA_ERR_HEAD_DIM
    assert lib.la_get_tile_sizes(128, 4, ctypes.byref(m), ctypes.byref(n)) == _cabi.LA_ERR_DTYPE
    for code in range(0, -14, -1):
        s = _cabi.status_string(code)
        assert s and s != "unknown la_status", code
    assert _cabi.status_string(-99) == "unknown la_status"


def test_argument_validation_returns_codes_without_launching():
    """Every check below fails before any HIP call, so it is safe without a GPU."""
    lib = _cabi.load()
    assert lib.la_fwd(None, None) == _cabi.LA_ERR_NULL_ARG
    a = _cabi.LaFwdArgs()
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_STRUCT_SIZE
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.dtype = 7                                                                # not one of LA_DTYPE_*
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_DTYPE
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == _cabi.LA_ERR_DTYPE
    a.dtype = _cabi.LA_DTYPE_FP16                                              # built: passes the dtype check like bf16
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_NULL_ARG
    a.dtype = _cabi.LA_DTYPE_BF16
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_NULL_ARG
    a.q = a.k = a.v = a.o = 0x1000
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_SHAPE
    a.batch, a.seqlen_q, a.seqlen_k, a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = 1, 256, 256, 4, 3, 128, 128
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_SHAPE           # heads_k must divide heads
    a.num_heads_k = 2                                                          # GQA is built (ABI 3): passes this check
    a.head_dim_v = 64
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_UNSUPPORTED     # head_dim_v != head_dim
    a.head_dim_v = 128
    a.reserved0 = 7
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_UNSUPPORTED     # reserved field must be 0
    a.reserved0 = 0
    a.flags = 0x84
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_UNSUPPORTED     # unknown flag
    a.flags = 0
    a.head_dim = a.head_dim_v = 100
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_HEAD_DIM        # % 8 (flash_api.cpp:854)
    a.head_dim = a.head_dim_v = 80
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_HEAD_DIM        # not instantiated (64 / 96 / 128 / 192 / 256 are)
    a.head_dim = a.head_dim_v = 96
    a.flags = _cabi.LA_FLAG_KERNEL_128ROW
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_HEAD_DIM        # the hipcc-scheduled A/B template has 64 / 128 / 256 only
    a.flags = 0
    a.head_dim = a.head_dim_v = 128
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_TILE_MISMATCH
    a.block_m, a.block_n = 256, 64
    a.read_list = 0x2000
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_LISTS
    a.write_list = 0x3000
    a.q_row_stride = 4 * 128 + 4
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_STRIDE
    # q-tile window (ABI 3): Sq = 256 with 256-row tiles is ONE q-tile
    a.q_row_stride = 4 * 128
    for begin, count in ((1, 0), (0, 2), (1, 1), (-1, 1), (0, -1)):
        a.q_tile_begin, a.q_tile_count = begin, count
        assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_Q_WINDOW, (begin, count)
    a.q_tile_begin = a.q_tile_count = 0
    # fp8: workspace contract
    a.q_row_stride = 4 * 128
    a.dtype = _cabi.LA_DTYPE_FP8_E4M3
    a.read_list = a.write_list = None
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == 1 * 2 * 4 * 8192 + 1024  # B * Hk * Kt * 8 KiB + ticket counter
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_WORKSPACE
    a.dtype = _cabi.LA_DTYPE_BF16
    a.block_m = 256
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == 1024                  # bf16 dense: the optional ticket counters too (round 5)
    a.flags = _cabi.LA_FLAG_STATIC_SCHED
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == 0                     # ... not under the static map
    a.flags = _cabi.LA_FLAG_KERNEL_128ROW
    a.block_m = 128
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == 0                     # ... nor for dense launches of the 128-row template
    a.flags, a.block_m = 0, 256
    a.read_list, a.write_list = 0x2000, 0x3000
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == 1024                  # bf16 with lists: the optional ticket counter
    assert lib.la_skip_list_stats(None, 1, 1, 1, 1, None, None) == _cabi.LA_ERR_NULL_ARG
    assert lib.la_combine(None, 0, None, None, _cabi.LA_DTYPE_BF16, None, 1, 1, 1, 1, 128, None) == _cabi.LA_ERR_NULL_ARG
    assert lib.la_combine(0x1000, 0, 0x1000, 0x1000, _cabi.LA_DTYPE_FP8_E4M3, None, 1, 1, 1, 1, 128, None) == _cabi.LA_ERR_DTYPE
    a.dtype = _cabi.LA_DTYPE_FP16
    assert lib.la_fwd_workspace_bytes(ctypes.byref(a)) == 1024                  # fp16 = bf16 here


def test_kernel_selection_is_an_argument_not_process_state():
    """ABI 4: the 128-row A/B kernel is chosen by LA_FLAG_KERNEL_128ROW (tile sizes follow through la_get_tile_sizes_ex);
    the C library reads no environment variable — the header promises no global state."""
    lib = _cabi.load()
    m, n = ctypes.c_int(), ctypes.c_int()
    assert lib.la_get_tile_sizes_ex(128, 2, 0, ctypes.byref(m), ctypes.byref(n)) == 0 and (m.value, n.value) == (256, 64)
    assert lib.la_get_tile_sizes_ex(128, 2, _cabi.LA_FLAG_KERNEL_128ROW, ctypes.byref(m), ctypes.byref(n)) == 0
    assert (m.value, n.value) == (128, 64)
    assert lib.la_get_tile_sizes_ex(64, 2, _cabi.LA_FLAG_KERNEL_128ROW, ctypes.byref(m), ctypes.byref(n)) == 0 and m.value == 128
    assert lib.la_get_tile_sizes_ex(128, 1, _cabi.LA_FLAG_KERNEL_128ROW, ctypes.byref(m), ctypes.byref(n)) == _cabi.LA_ERR_UNSUPPORTED
    assert lib.la_get_tile_sizes_ex(128, 2, 0x100, ctypes.byref(m), ctypes.byref(n)) == _cabi.LA_ERR_UNSUPPORTED
    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.q = a.k = a.v = a.o = 0x1000
    a.batch, a.seqlen_q, a.seqlen_k, a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = 1, 256, 256, 4, 4, 128, 128
    a.block_m, a.block_n = 256, 64
    a.flags = _cabi.LA_FLAG_KERNEL_128ROW
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_TILE_MISMATCH       # lists of this launch use 128-row q-tiles
    csrc = os.path.join(ROOT, "liteattention_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    env = dict(os.environ, LA_FWD_KERNEL="v2")                                    # the HOST layer's default flags follow the env
    code = ("import sys; sys.path.insert(0, %r)\nfrom liteattention_amd import _cabi\n"
            "assert _cabi.default_flags() == _cabi.LA_FLAG_KERNEL_128ROW and _cabi.get_tile_sizes(128, 2) == (128, 64)\n"
            "assert _cabi.get_tile_sizes(128, 1) == (256, 64)\n" % ROOT)
    assert subprocess.run([sys.executable, "-c", code], env=env).returncode == 0


def test_varlen_argument_validation():
    """cu_seqlens (ABI 4): both or neither; skip lists only where built; no q-tile window. All rejected before any HIP call."""
    lib = _cabi.load()
    a = _cabi.LaFwdArgs()
    a.struct_size = ctypes.sizeof(_cabi.LaFwdArgs)
    a.q = a.k = a.v = a.o = 0x1000
    a.batch, a.seqlen_q, a.seqlen_k, a.num_heads, a.num_heads_k, a.head_dim, a.head_dim_v = 3, 300, 512, 4, 4, 128, 128
    a.block_m, a.block_n = 256, 64
    a.q_row_stride = a.k_row_stride = a.v_row_stride = a.o_row_stride = 4 * 128
    a.q_head_stride = a.k_head_stride = a.v_head_stride = a.o_head_stride = 128
    a.cu_seqlens_q = 0x4000
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_NULL_ARG            # cu_seqlens_k missing
    a.cu_seqlens_k = 0x5000
    a.total_q = 889
    a.read_list, a.write_list = 0x2000, 0x3000                                    # skip lists with varlen: head_dim <= 128 on the
    a.flags = _cabi.LA_FLAG_KERNEL_128ROW                                         # hand-scheduled kernels only (round 3)
    a.block_m = 128
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_UNSUPPORTED
    a.flags, a.block_m = 0, 128
    a.head_dim = a.head_dim_v = 256
    a.q_row_stride = a.k_row_stride = a.v_row_stride = a.o_row_stride = 4 * 256
    a.q_head_stride = a.k_head_stride = a.v_head_stride = a.o_head_stride = 256
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_UNSUPPORTED
    a.head_dim = a.head_dim_v = 128
    a.block_m = 256
    a.q_row_stride = a.k_row_stride = a.v_row_stride = a.o_row_stride = 4 * 128
    a.q_head_stride = a.k_head_stride = a.v_head_stride = a.o_head_stride = 128
    a.read_list = a.write_list = None
    a.q_tile_count = 1
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_SHAPE               # no q-tile windows with varlen
    a.q_tile_count = 0
    a.total_q = -1
    assert lib.la_fwd(ctypes.byref(a), None) == _cabi.LA_ERR_SHAPE


def test_missing_library_fails_loudly(tmp_path):
    """The product must not fall back to anything when the HIP extension is absent: import raises."""
    code = "import sys; sys.path.insert(0, %r)\nimport liteattention_amd\n" % ROOT
    env = dict(os.environ, LITEATTENTION_AMD_LIB=str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode != 0
    assert "NativeLibraryError" in out.stderr and "no CPU/eager fallback" in out.stderr


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no product source may import, include or load it."""
    pkg = os.path.join(ROOT, "liteattention_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#include\s+.*oracle)|qkskip_oracle|libqkskip", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), f


# ------------------------------------------------------------------------------------------ the documents say only true things
DOCS = ("include/lite_attention_amd.h", "DESIGN.md", "INTEGRATION.md", "README.md")
CSRC = os.path.join(ROOT, "liteattention_amd", "csrc")


def _nm_exports():
    """Dynamic symbols `la_*` of the built library (nm -D): the entry points a non-Python host can bind."""
    out = subprocess.run(["nm", "-D", "--defined-only", _cabi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    return sorted({ln.split()[-1] for ln in out.splitlines() if re.fullmatch(r"la_[a-z0-9_]+", ln.split()[-1])})


def _header_type_names():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return set(re.findall(r"\b(?:struct|enum)\s+(la_\w+)", src)) | set(re.findall(r"}\s*(la_\w+)\s*;", src))


def test_every_la_name_in_the_documents_exists():
    """ABI 5 advertised `la_blockmask_to_lists` in the header comment, DESIGN.md and INTEGRATION.md while no
    source defined it. Every `la_*` token of the boundary documents must be (a) a dynamic symbol of the built library, (b) a
    struct / enum / typedef of the header, (c) a source file of csrc/ (token == file stem), or (d) a device kernel defined there."""
    exports, types = set(_nm_exports()), _header_type_names()
    stems = {os.path.splitext(f)[0] for f in os.listdir(CSRC)}
    kernel_src = "\n".join(open(os.path.join(CSRC, f)).read() for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    bad = []
    for doc in DOCS:
        for tok in sorted(set(re.findall(r"\bla_[a-z0-9_]+\b", open(os.path.join(ROOT, doc)).read()))):
            if tok in exports or tok in types or tok in stems:
                continue
            if re.search(r"__global__[^;{]*\b" + re.escape(tok) + r"\s*\(", kernel_src):
                continue
            bad.append((doc, tok))
    assert not bad, f"names the documents mention that exist nowhere: {bad}"


def test_every_exported_symbol_is_declared_bound_and_documented():
    """... and vice versa: what `nm -D` shows is exactly what the header declares and the ctypes layer binds, and INTEGRATION.md
    names each entry point."""
    exports = _nm_exports()
    assert exports == declared_functions() == sorted(_cabi.EXPORTED_SYMBOLS)
    integration = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in exports:
        assert re.search(r"\b" + name + r"\b", integration), f"INTEGRATION.md never mentions {name}"


def test_round4_entry_points_validate_without_a_device():
    """la_blockmask_to_lists / la_device_slots (ABI 6): the argument checks fail before any HIP call (safe without a GPU); the slot query
    answers from the tile table + the device attribute (256 compute units when there is no device to ask)."""
    lib = _cabi.load()
    buf = (ctypes.c_int32 * 16)()
    m = (ctypes.c_uint8 * 16)()
    assert lib.la_blockmask_to_lists(None, 0, 0, 1, 1, 1, 1, None, None, buf, None, None) == _cabi.LA_ERR_NULL_ARG
    assert lib.la_blockmask_to_lists(m, 0, 0, 1, 1, 1, 1, None, None, None, None, None) == _cabi.LA_ERR_NULL_ARG
    assert lib.la_blockmask_to_lists(m, 0, 0, 0, 1, 1, 1, None, None, buf, None, None) == _cabi.LA_ERR_SHAPE
    assert lib.la_blockmask_to_lists(m, 0, 0, 1, 1, 1, 0, None, None, buf, None, None) == _cabi.LA_ERR_SHAPE
    assert lib.la_blockmask_to_lists(m, 0, -4, 1, 1, 1, 1, None, None, buf, None, None) == _cabi.LA_ERR_STRIDE
    cu, per = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.la_device_slots(128, 2, 0, ctypes.byref(cu), ctypes.byref(per)) == _cabi.LA_OK and cu.value > 0 and per.value == 1
    assert lib.la_device_slots(64, 2, 0, ctypes.byref(cu), ctypes.byref(per)) == _cabi.LA_OK and per.value == 1       # the one-wave body is the default at head_dim 64
    assert lib.la_device_slots(128, 2, _cabi.LA_FLAG_KERNEL_128ROW, ctypes.byref(cu), ctypes.byref(per)) == _cabi.LA_OK and per.value == 2
    assert lib.la_device_slots(256, 2, _cabi.LA_FLAG_KERNEL_128ROW, ctypes.byref(cu), ctypes.byref(per)) == _cabi.LA_OK and per.value == 1
    assert lib.la_device_slots(48, 2, 0, ctypes.byref(cu), ctypes.byref(per)) == _cabi.LA_ERR_HEAD_DIM
    assert lib.la_device_slots(128, 1, _cabi.LA_FLAG_KERNEL_128ROW, None, None) == _cabi.LA_ERR_UNSUPPORTED
    assert _cabi.device_slots(128, 2)[1] == 1


def test_window_planning_maps_the_head_dim_before_asking_the_library():
    """``HeadShardedLiteAttention.q_windows`` asked ``la_device_slots`` with the RAW head dim / element size, which
    the library only answers for instantiated kernels (80, 72 -> LA_ERR_HEAD_DIM), so every overlapped call at a zero-padded head dim
    raised. The host maps exactly as ``get_tile_sizes`` does (80 -> 96); e4m3 at 192 / 256 is answered by the library itself since round 6
    (la_fwd serves it with the bf16 kernel of that head dim: its tiles, its slots)."""
    from liteattention_amd import flash_attn_interface as fai
    lib = _cabi.load()
    cu, per = ctypes.c_int(), ctypes.c_int()
    for d, e in ((80, 2), (72, 2), (160, 1)):
        assert lib.la_device_slots(d, e, 0, ctypes.byref(cu), ctypes.byref(per)) == _cabi.LA_ERR_HEAD_DIM      # the raw question fails
    m, n = ctypes.c_int(), ctypes.c_int()
    assert lib.la_get_tile_sizes(256, 1, ctypes.byref(m), ctypes.byref(n)) == _cabi.LA_OK and (m.value, n.value) == (128, 64)
    assert lib.la_get_tile_sizes(192, 1, ctypes.byref(m), ctypes.byref(n)) == _cabi.LA_OK and (m.value, n.value) == (128, 64)
    try:
        want96, want256 = _cabi.device_slots(96, 2), _cabi.device_slots(256, 2)
    except RuntimeError:
        pytest.skip("la_device_slots needs a device for the compute-unit count")
    assert fai.device_slots(80, 2) == fai.device_slots(72, 2) == want96
    assert fai.device_slots(256, 1) == want256 and fai.device_slots(160, 1) == _cabi.device_slots(192, 2)
