"""ctypes binding of the C-ABI in include/lite_attention_amd.h.

This is the only place the package touches native code. It NEVER falls back to a CPU or eager
implementation: if the shared library is missing or a symbol is absent, import-time loading raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Tuple

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# LITEATTENTION_AMD_LIB overrides the in-tree location (deployment / tests of the failure path)
LIB_PATH = os.environ.get("LITEATTENTION_AMD_LIB") or os.path.join(_PKG_DIR, "libliteattention_amd.so")

LA_ABI_VERSION = 8
LA_DTYPE_BF16, LA_DTYPE_FP16, LA_DTYPE_FP8_E4M3, LA_DTYPE_FP32 = 0, 1, 2, 3

LA_OK = 0
LA_ERR_NULL_ARG, LA_ERR_STRUCT_SIZE, LA_ERR_DTYPE, LA_ERR_HEAD_DIM, LA_ERR_SHAPE = -1, -2, -3, -4, -5
LA_ERR_STRIDE, LA_ERR_TILE_MISMATCH, LA_ERR_LISTS, LA_ERR_UNSUPPORTED, LA_ERR_LAUNCH, LA_ERR_SEQLEN = (
    -6, -7, -8, -9, -10, -11)
LA_ERR_WORKSPACE = -12
LA_ERR_Q_WINDOW = -13

EXPORTED_SYMBOLS = (
    "la_abi_version", "la_get_tile_sizes", "la_get_tile_sizes_ex", "la_fwd", "la_fwd_workspace_bytes", "la_skip_list_stats", "la_combine",
    "la_status_string", "la_last_hip_error", "la_blockmask_to_lists", "la_device_slots", "la_build_info", "la_combine_list",
)


class LaFwdArgs(ctypes.Structure):
    """Mirror of ``la_fwd_args`` (field order and types must match the header)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("dtype", ctypes.c_int32),
        ("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p),
        ("o", ctypes.c_void_p), ("lse", ctypes.c_void_p),
        ("q_batch_stride", ctypes.c_int64), ("q_row_stride", ctypes.c_int64), ("q_head_stride", ctypes.c_int64),
        ("k_batch_stride", ctypes.c_int64), ("k_row_stride", ctypes.c_int64), ("k_head_stride", ctypes.c_int64),
        ("v_batch_stride", ctypes.c_int64), ("v_row_stride", ctypes.c_int64), ("v_head_stride", ctypes.c_int64),
        ("o_batch_stride", ctypes.c_int64), ("o_row_stride", ctypes.c_int64), ("o_head_stride", ctypes.c_int64),
        ("batch", ctypes.c_int32), ("seqlen_q", ctypes.c_int32), ("seqlen_k", ctypes.c_int32),
        ("num_heads", ctypes.c_int32), ("num_heads_k", ctypes.c_int32),
        ("head_dim", ctypes.c_int32), ("head_dim_v", ctypes.c_int32),
        ("softmax_scale", ctypes.c_float),
        ("q_descale", ctypes.c_void_p), ("k_descale", ctypes.c_void_p), ("v_descale", ctypes.c_void_p),
        ("q_descale_batch_stride", ctypes.c_int64), ("q_descale_head_stride", ctypes.c_int64),
        ("k_descale_batch_stride", ctypes.c_int64), ("k_descale_head_stride", ctypes.c_int64),
        ("v_descale_batch_stride", ctypes.c_int64), ("v_descale_head_stride", ctypes.c_int64),
        ("read_list", ctypes.c_void_p), ("write_list", ctypes.c_void_p), ("must_do_list", ctypes.c_void_p),
        ("must_do_is_1d", ctypes.c_int32), ("thr", ctypes.c_float),
        ("block_m", ctypes.c_int32), ("block_n", ctypes.c_int32),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64),
        ("q_tile_begin", ctypes.c_int32), ("q_tile_count", ctypes.c_int32),
        ("flags", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
        ("cu_seqlens_q", ctypes.c_void_p), ("cu_seqlens_k", ctypes.c_void_p), ("total_q", ctypes.c_int64),
    ]


LA_FLAG_V_PREPARED = 1
LA_FLAG_STATIC_SCHED = 2
LA_FLAG_KERNEL_128ROW = 4
LA_FLAG_EXACT_RESCALE = 8
LA_FLAG_FP8_MFMA_ROWSUM = 16     # fp8: row sums of the ROUNDED P from the matrix pipe (default: fp32 sums of the un-rounded P, the reference's)
LA_FLAG_FP8_ENCODED_P = 32       # fp8: the block-scaled log-linear byte encoding of P (default: exp2 + hardware e4m3 rounding, the reference's)
LA_FLAG_HALF_VOTE = 64
GEOMETRY_FLAGS = LA_FLAG_KERNEL_128ROW | LA_FLAG_HALF_VOTE      # the flags that change the q-tile of the skip lists (la_get_tile_sizes_ex)


def default_flags() -> int:
    """A/B switches of the HOST layer (the C library reads no environment): they only choose the default ``la_fwd_args.flags``.
    LA_FWD_KERNEL=v2 -> the 128-row bf16 head_dim-128 kernel (lists then use 128-row q-tiles); LA_VOTE=half -> LA_FLAG_HALF_VOTE (the
    hand-scheduled head_dim-128 kernel with lists per 128-row half); LA_SCHED=static -> one
    workgroup per item instead of the ticket queues; LA_RESCALE_TAU=0 -> O rescaled on every growth of a row maximum; LA_FP8_P selects
    the fp8 form of P: unset / "reference" -> the reference's arithmetic (the default since round 6: exp2 + hardware e4m3 rounding, fp32 row sums of the
    un-rounded P), "mfma_rowsum" -> LA_FLAG_FP8_MFMA_ROWSUM (row sums of the rounded P from the matrix pipe), "encoded" -> LA_FLAG_FP8_ENCODED_P (the
    block-scaled log-linear byte encoding: the fast form, NOT the reference's arithmetic)."""
    f = 0
    if os.environ.get("LA_FWD_KERNEL", "").startswith("v2"):
        f |= LA_FLAG_KERNEL_128ROW
    if os.environ.get("LA_VOTE", "").startswith("half"):        # skip lists per 128-row half of the 256-row workgroup (bf16 / fp16 head_dim 128)
        f |= LA_FLAG_HALF_VOTE
    if os.environ.get("LA_SCHED", "").startswith("s"):
        f |= LA_FLAG_STATIC_SCHED
    if os.environ.get("LA_RESCALE_TAU", "") not in ("", "8", "8.0"):
        if float(os.environ["LA_RESCALE_TAU"]) != 0.0:
            raise ValueError("LA_RESCALE_TAU: only 0 (exact rescale, LA_FLAG_EXACT_RESCALE) or the default 8 are available")
        f |= LA_FLAG_EXACT_RESCALE
    fp8_p = os.environ.get("LA_FP8_P", "")
    if fp8_p not in ("", "reference", "mfma_rowsum", "encoded"):
        raise ValueError("LA_FP8_P: reference (default), mfma_rowsum or encoded")
    if fp8_p == "mfma_rowsum":
        f |= LA_FLAG_FP8_MFMA_ROWSUM
    if fp8_p == "encoded":
        f |= LA_FLAG_FP8_ENCODED_P
    return f


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """dlopen the HIP extension. Raises NativeLibraryError (never falls back)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; this library NEEDs the same SONAME. Whichever is
    # loaded first serves both, and a process must have exactly one HIP runtime (with the system one
    # loaded first, torch's device init reports hipErrorNoDevice). So: torch first, always.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m liteattention_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for this op.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name in EXPORTED_SYMBOLS:
        if not hasattr(lib, name):
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}")
    lib.la_abi_version.restype = ctypes.c_int
    lib.la_get_tile_sizes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.la_get_tile_sizes.restype = ctypes.c_int
    lib.la_get_tile_sizes_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int),
                                         ctypes.POINTER(ctypes.c_int)]
    lib.la_get_tile_sizes_ex.restype = ctypes.c_int
    lib.la_fwd.argtypes = [ctypes.POINTER(LaFwdArgs), ctypes.c_void_p]
    lib.la_fwd.restype = ctypes.c_int
    lib.la_fwd_workspace_bytes.argtypes = [ctypes.POINTER(LaFwdArgs)]
    lib.la_fwd_workspace_bytes.restype = ctypes.c_int64
    lib.la_skip_list_stats.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.la_skip_list_stats.restype = ctypes.c_int
    lib.la_combine.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                               ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                               ctypes.c_void_p]
    lib.la_combine.restype = ctypes.c_int
    lib.la_combine_list.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int32,
                                    ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.la_combine_list.restype = ctypes.c_int
    lib.la_blockmask_to_lists.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.la_blockmask_to_lists.restype = ctypes.c_int
    lib.la_device_slots.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.la_device_slots.restype = ctypes.c_int
    lib.la_status_string.argtypes = [ctypes.c_int]
    lib.la_status_string.restype = ctypes.c_char_p
    lib.la_last_hip_error.restype = ctypes.c_int
    lib.la_build_info.restype = ctypes.c_char_p
    if lib.la_abi_version() != LA_ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {lib.la_abi_version()} vs binding {LA_ABI_VERSION}")
    _check_build_record(lib)
    _lib = lib
    return lib


def build_info() -> dict:
    """``la_build_info()`` of the loaded library as a dict: abi, src (hash of the sources it was built from), variant, wrong_results, opts."""
    from . import _buildinfo
    return _buildinfo.parse(load().la_build_info().decode())


def _check_build_record(lib) -> None:
    """The library in the DEFAULT location must be the product build of the sources beside it: generated with no generator option and
    no define (variant=0, wrong_results=0) from exactly this tree (src). A/B and pricing variants - whose results may be wrong on
    purpose - are only reachable by naming them in LITEATTENTION_AMD_LIB, which skips this check (the caller chose the file)."""
    if os.environ.get("LITEATTENTION_AMD_LIB"):
        return
    from . import _buildinfo
    rec = _buildinfo.parse(lib.la_build_info().decode())
    if rec.get("variant") != "0" or rec.get("wrong_results") != "0":
        raise NativeLibraryError(f"{LIB_PATH} is an A/B or pricing variant ({lib.la_build_info().decode()}), not the product build: "
                                 "rebuild with `python -m liteattention_amd.build --force`")
    want = _buildinfo.source_hash()
    if want is not None and rec.get("src") != want:
        raise NativeLibraryError(f"{LIB_PATH} was built from other sources (library src={rec.get('src')}, tree src={want}): "
                                 "rebuild with `python -m liteattention_amd.build`")


def status_string(code: int) -> str:
    return load().la_status_string(code).decode()


def is_instantiated(head_dim: int, element_size: int, flags: int = None) -> bool:
    """Does la_fwd have a kernel for exactly this head_dim (under the kernel-selection flags)? The library is the one table."""
    m, n = ctypes.c_int(0), ctypes.c_int(0)
    f = 0 if element_size == 1 else (default_flags() if flags is None else flags) & GEOMETRY_FLAGS
    return load().la_get_tile_sizes_ex(int(head_dim), int(element_size), f, ctypes.byref(m), ctypes.byref(n)) == LA_OK


_TILE_SIZES = {}      # (head_dim, element_size, kernel-selection flag) -> (block_m, block_n): constant for a loaded library


def get_tile_sizes(head_dim: int, element_size: int, flags: int = None) -> Tuple[int, int]:
    """(kBlockM, kBlockN) of the kernel la_fwd runs for this head_dim / element size (and kernel-selection flags; default:
    ``default_flags()``, what ``mha_fwd`` passes). fp8 has no 128-row kernel: the flag is dropped for 1-byte elements."""
    m, n = ctypes.c_int(0), ctypes.c_int(0)
    f = (default_flags() if flags is None else flags) & GEOMETRY_FLAGS
    if element_size == 1:
        f = 0
    key = (int(head_dim), int(element_size), f)
    hit = _TILE_SIZES.get(key)
    if hit is not None:
        return hit
    rc = load().la_get_tile_sizes_ex(int(head_dim), int(element_size), f, ctypes.byref(m), ctypes.byref(n))
    if rc != LA_OK:
        raise RuntimeError(f"la_get_tile_sizes(head_dim={head_dim}, element_size={element_size}): {status_string(rc)}")
    _TILE_SIZES[key] = (m.value, n.value)
    return m.value, n.value


def device_slots(head_dim: int, element_size: int, flags: int = None) -> Tuple[int, int]:
    """(compute units of the current device, resident workgroups per compute unit) for the kernel la_fwd runs: what a host that
    issues one attention as several q-tile windows sizes its windows with (``parallel.plan_q_windows``)."""
    cu, per = ctypes.c_int(0), ctypes.c_int(0)
    f = (default_flags() if flags is None else flags) & GEOMETRY_FLAGS
    if element_size == 1:
        f = 0
    rc = load().la_device_slots(int(head_dim), int(element_size), f, ctypes.byref(cu), ctypes.byref(per))
    if rc != LA_OK:
        raise RuntimeError(f"la_device_slots(head_dim={head_dim}, element_size={element_size}): {status_string(rc)}")
    return cu.value, per.value
