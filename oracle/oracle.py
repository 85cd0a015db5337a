"""CPU oracle for the QK-Skip attention hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module. The product package (``liteattention_amd``) never does; it fails loudly
when its HIP extension is missing.

Contents
--------
* ``qkskip_fwd``            ctypes front-end of ``oracle/qkskip_oracle.c`` (tiled walk, skip lists).
* ``attention_dense_ref``   eager dense oracle; restates the non-causal subset of
                            ``attention_ref`` (/root/reference/hopper/tests/test_util.py:226-348)
                            and additionally returns the log-sum-exp
                            (/root/reference/test_lite_attention.py:67-77).
* ``attention_combine_ref`` restates ``attention_combine_ref``
                            (/root/reference/hopper/tests/test_flash_attn.py:1178-1187).
* pure-Python restatements of the host bookkeeping (``init_skip_list_ref``,
  ``expand_must_do_ref``, ``simulate_writer``, ``listed_tiles``) following
  /root/reference/hopper/lite_attention.py:113-153, 214-242 and
  /root/reference/hopper/_internal/cpp/mainloop_fwd_sm90_tma_gmma_ws.hpp:47-192, 1804-1827.

Parity status: dense O/LSE pinned by tests/golden/dense_*.npz (reference outputs); host
bookkeeping pinned by tests/golden/host_*.json (reference outputs); skip-list contents beyond
the reference's behavioural checks K1-K4 are "parity unpinned" by the reference (see the
header of qkskip_oracle.c).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libqkskip_oracle.so")
_lib = None


class _Args(ctypes.Structure):
    _fields_ = [
        ("q", ctypes.c_void_p), ("k", ctypes.c_void_p), ("v", ctypes.c_void_p),
        ("o", ctypes.c_void_p), ("lse", ctypes.c_void_p),
        ("B", ctypes.c_int32), ("Sq", ctypes.c_int32), ("Sk", ctypes.c_int32),
        ("H", ctypes.c_int32), ("D", ctypes.c_int32), ("Dv", ctypes.c_int32),
        ("softmax_scale", ctypes.c_float),
        ("block_m", ctypes.c_int32), ("block_n", ctypes.c_int32),
        ("read_list", ctypes.c_void_p), ("write_list", ctypes.c_void_p),
        ("must_do_list", ctypes.c_void_p), ("must_do_is_1d", ctypes.c_int32),
        ("thr", ctypes.c_float),
        ("p_round", ctypes.c_int32), ("mask_any_tail", ctypes.c_int32),
        ("nthreads", ctypes.c_int32),
        ("tiles_done", ctypes.c_void_p),
        ("q_descale", ctypes.c_void_p), ("k_descale", ctypes.c_void_p), ("v_descale", ctypes.c_void_p),
        ("margins", ctypes.c_void_p),
        ("lin_tau", ctypes.c_float), ("lin_group", ctypes.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/qkskip_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "qkskip_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load_lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.la_oracle_fwd.argtypes = [ctypes.POINTER(_Args)]
        _lib.la_oracle_fwd.restype = ctypes.c_int
        _lib.la_oracle_struct_size.restype = ctypes.c_int
        assert _lib.la_oracle_struct_size() == ctypes.sizeof(_Args), "oracle struct mismatch"
    return _lib


def _f32c(x: torch.Tensor) -> torch.Tensor:
    return x.detach().to("cpu", torch.float32).contiguous()


def qkskip_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, *, block_m: int, block_n: int,
               read_list: Optional[torch.Tensor] = None, write_list: Optional[torch.Tensor] = None,
               must_do_list: Optional[torch.Tensor] = None, thr: float = -3.0,
               softmax_scale: Optional[float] = None, p_round: bool = True,
               mask_any_tail: bool = True, nthreads: int = 0, margins: Optional[torch.Tensor] = None,
               q_descale: Optional[torch.Tensor] = None, k_descale: Optional[torch.Tensor] = None,
               v_descale: Optional[torch.Tensor] = None, lin_lazy: bool = True) -> Tuple[torch.Tensor, torch.Tensor, int]:
    """Tiled CPU forward. q,k,v: (B,S,H,D) of any float dtype (values are taken as they are,
    i.e. bf16 tensors give bf16-representable fp32 operands). Lists are CPU int32 tensors of shape
    [>=B, H, Qt, Kt+1]; ``write_list`` is filled in place. ``must_do_list`` may be 1-D ([Kt+1]).
    ``margins`` (optional fp32 [B,H,Qt,Kt]) receives, per computed non-first tile, the quantity compared
    with ``thr`` (skip <=> margin <= thr); NaN elsewhere.
    Returns (o fp32 (B,Sq,H,Dv), lse fp32 (B,H,Sq), number of K tiles computed)."""
    lib = load_lib()
    qf, kf, vf = _f32c(q), _f32c(k), _f32c(v)
    B, Sq, H, D = qf.shape
    Sk, Dv = kf.shape[1], vf.shape[3]
    if kf.shape[2] != H:
        # GQA / MQA: query head h uses K/V head h // g — restated as the reference's oracle does it, by repeating the
        # K/V heads (and the per-K/V-head descales) g times (hopper/tests/test_util.py:275-284)
        g = H // kf.shape[2]
        assert kf.shape[2] * g == H and vf.shape[2] == kf.shape[2], "nheads_k must divide nheads"
        kf, vf = kf.repeat_interleave(g, dim=2).contiguous(), vf.repeat_interleave(g, dim=2).contiguous()
        q_descale, k_descale, v_descale = [None if t is None else t.detach().cpu().repeat_interleave(g, dim=1)
                                           for t in (q_descale, k_descale, v_descale)]
    assert kf.shape == (B, Sk, H, D) and vf.shape == (B, Sk, H, Dv)
    if softmax_scale is None:
        softmax_scale = D ** -0.5
    o = torch.empty(B, Sq, H, Dv, dtype=torch.float32)
    lse = torch.empty(B, H, Sq, dtype=torch.float32)
    tiles = ctypes.c_int64(0)
    Qt, Kt = -(-Sq // block_m), -(-Sk // block_n)

    def _chk(t, name):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu", name
        assert t.dim() == 4 and t.shape[0] >= B and tuple(t.shape[1:]) == (H, Qt, Kt + 1), (name, t.shape)
        return t.data_ptr()

    a = _Args()
    a.q, a.k, a.v, a.o, a.lse = qf.data_ptr(), kf.data_ptr(), vf.data_ptr(), o.data_ptr(), lse.data_ptr()
    a.B, a.Sq, a.Sk, a.H, a.D, a.Dv = B, Sq, Sk, H, D, Dv
    a.softmax_scale = softmax_scale
    a.block_m, a.block_n = block_m, block_n
    a.read_list = _chk(read_list, "read_list") if read_list is not None else None
    a.write_list = _chk(write_list, "write_list") if write_list is not None else None
    if must_do_list is not None:
        assert must_do_list.dtype == torch.int32 and must_do_list.is_contiguous()
        if must_do_list.dim() == 1:
            assert must_do_list.numel() >= 3
            a.must_do_is_1d = 1
            a.must_do_list = must_do_list.data_ptr()
        else:
            a.must_do_is_1d = 0
            a.must_do_list = _chk(must_do_list, "must_do_list")
    a.thr = thr
    # "fp8": e4m3 P with the 2^8 offset (reference Max_offset); "f16": P rounded to fp16; True / False: bf16 / fp32 P;
    # "fp8_lin": NOT a reference form - this build's default log-linear byte encoding of P (qkskip_oracle.c, p_round 4)
    a.p_round = {"fp8": 2, "f16": 3, "fp8_lin": 4}.get(p_round, None) if isinstance(p_round, str) else int(p_round)
    # "fp8_lin": by default the kernel's own LAZY reference maximum (m_ref = the first walked tile's row maximum; a wave of 64 rows moves it
    # only after growth by more than 32 log2 units), so that the encoded bytes are the kernel's wherever the scores agree exactly;
    # lin_lazy=False: reference = the true running maximum after every tile (the restatement of rounds 3-4; another grid of the same kind)
    a.lin_tau, a.lin_group = (32.0, 64) if (a.p_round == 4 and lin_lazy) else (0.0, 0)
    keep = []
    for name, t in (("q_descale", q_descale), ("k_descale", k_descale), ("v_descale", v_descale)):
        if t is not None:
            t32 = t.detach().to("cpu", torch.float32).contiguous()
            assert tuple(t32.shape) == (B, H), name
            keep.append(t32)
            setattr(a, name, t32.data_ptr())
    a.mask_any_tail = int(mask_any_tail)
    a.nthreads = nthreads
    a.tiles_done = ctypes.addressof(tiles)
    if margins is not None:
        assert margins.dtype == torch.float32 and margins.is_contiguous() and tuple(margins.shape) == (B, H, Qt, Kt)
        margins.fill_(float("nan"))
        a.margins = margins.data_ptr()
    rc = lib.la_oracle_fwd(ctypes.byref(a))
    if rc != 0:
        raise RuntimeError(f"la_oracle_fwd failed: {rc}")
    return o, lse, int(tiles.value)


def attention_dense_ref(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                        softmax_scale: Optional[float] = None, upcast: bool = True,
                        reorder_ops: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eager dense attention (non-causal subset of test_util.py:226-348).

    upcast=True  -> the fp32 "out_ref"; upcast=False, reorder_ops=True -> the same-dtype "out_pt"
    used by the reference tolerance rule (test_flash_attn.py:266-296). Returns (out in q.dtype,
    lse fp32 (B,H,Sq))."""
    dtype_og = q.dtype
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
    if k.shape[2] != q.shape[2]:                       # GQA / MQA (test_util.py:283-284)
        g = q.shape[2] // k.shape[2]
        k, v = k.repeat_interleave(g, dim=2), v.repeat_interleave(g, dim=2)
    d = q.shape[-1]
    if softmax_scale is None:
        softmax_scale = 1.0 / math.sqrt(d)
    if not reorder_ops:
        scores = torch.einsum("bthd,bshd->bhts", q * softmax_scale, k)
    else:
        scores = torch.einsum("bthd,bshd->bhts", q, k * softmax_scale)
    lse = torch.logsumexp(scores.float(), dim=-1)
    attention = torch.softmax(scores, dim=-1).to(v.dtype)
    out = torch.einsum("bhts,bshd->bthd", attention, v)
    return out.to(dtype_og), lse


def attention_dense_ref_chunked(q, k, v, softmax_scale=None, chunk: int = 2048):
    """fp32 dense oracle evaluated in query chunks (bounded memory for long sequences)."""
    outs, lses = [], []
    for s in range(0, q.shape[1], chunk):
        o, l = attention_dense_ref(q[:, s:s + chunk], k, v, softmax_scale)
        outs.append(o)
        lses.append(l)
    return torch.cat(outs, dim=1), torch.cat(lses, dim=2)


def round_like_p(x: torch.Tensor, p_round) -> torch.Tensor:
    """The C oracle's rounding of P applied to a float32 tensor (p_round as in qkskip_fwd: True/"bf16", "fp8", "f16";
    "fp8_lin": x is taken as 8 y + 56 - 8 delta and mapped to the e4m3 value of the byte the kernel would store)."""
    y = x.detach().to(torch.float32).contiguous().clone()
    mode = {"fp8": 2, "f16": 3, "fp8_lin": 4}[p_round] if isinstance(p_round, str) else int(bool(p_round))
    lib = load_lib()
    lib.la_oracle_round.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    lib.la_oracle_round.restype = None
    lib.la_oracle_round(y.data_ptr(), y.numel(), mode)
    return y


def attention_combine_ref(out_partial: torch.Tensor, lse_partial: torch.Tensor):
    """out_partial (splits,B,S,H,D), lse_partial (splits,B,S,H)  -> (out, lse (B,S,H)).
    Follows test_flash_attn.py:1178-1187."""
    lse = torch.logsumexp(lse_partial, dim=0)
    scale = torch.exp(lse_partial - lse)
    scale = torch.where(torch.isinf(scale) | torch.isnan(scale), torch.zeros_like(scale), scale)
    out = (scale.unsqueeze(-1) * out_partial).sum(0)
    return out, lse


# --------------------------------------------------------------------------------------------
# host bookkeeping restatements (small, pure Python)
# --------------------------------------------------------------------------------------------

def ceil_div(x: int, y: int) -> int:
    return (x + y - 1) // y


def init_skip_list_ref(batch: int, q_tiles: int, k_tiles: int, heads: int) -> torch.Tensor:
    """lite_attention.py:113-153 (must_skip_list=None branch): [2,B,H,Qt,Kt+1], row [2,Kt-1,0,...]."""
    sl = torch.zeros(2, batch, heads, q_tiles, k_tiles + 1, dtype=torch.int32)
    sl[..., 1] = k_tiles - 1
    sl[..., 0] = 2
    return sl


def expand_must_do_ref(must_do_list: Sequence[int], k_tile: int, width: int) -> torch.Tensor:
    """lite_attention.py:214-242 without the 4-D repeat: 1-D int32 row of length `width`."""
    lst = [len(must_do_list)] + list(must_do_list)
    for i in range(1, lst[0] + 1):
        if i % 2 == 1:
            lst[i] = (lst[i] + k_tile - 1) // k_tile
        else:
            lst[i] = lst[i] // k_tile
    row = torch.zeros(width, dtype=torch.int32)
    row[: len(lst)] = torch.tensor(lst, dtype=torch.int32)
    return row


def blockmask_rows_ref(mask2d, k_tiles_valid: Optional[int] = None) -> torch.Tensor:
    """Checker of ``la_blockmask_to_lists``: a 0/1 block mask [q_tiles, k_tiles] -> int32 rows [q_tiles, k_tiles + 1] in the list format the
    reader walks (row = [L, start0, end0, ...], ranges descending, both ends inclusive: mainloop_fwd_sm90_tma_gmma_ws.hpp:47-115),
    zero padded; an entry that would lie behind the row is counted in L and not stored (the reader takes a missing end as 0); a row that
    keeps nothing is [0, 0, ...]. Pure Python, one tile at a time; the inverse is ``walk_tiles``."""
    qt, kt = int(mask2d.shape[0]), int(mask2d.shape[1])
    kv = kt if k_tiles_valid is None else max(0, min(int(k_tiles_valid), kt))
    out = torch.zeros(qt, kt + 1, dtype=torch.int32)
    for m in range(qt):
        keep = [bool(mask2d[m, t]) and t < kv for t in range(kt)]
        row, t = [], kt - 1
        while t >= 0:
            if keep[t]:
                start = t
                while t - 1 >= 0 and keep[t - 1]:
                    t -= 1
                row += [start, t]
            t -= 1
        out[m, 0] = len(row)
        for i, x in enumerate(row):
            if 1 + i <= kt:
                out[m, 1 + i] = x
    return out


def walk_tiles(row: Sequence[int]) -> List[int]:
    """Tile indices the reader visits for one list row, in visiting order
    (mainloop...:1804-1827; both range ends inclusive)."""
    L = int(row[0])
    out: List[int] = []
    idx = 1

    def at(i):          # an entry behind the row reads as 0 (qkskip_oracle.c reader_load: a one-tile row [len, start] has no room for its end)
        return int(row[i]) if i < len(row) else 0
    start, end = at(1), at(2)
    while True:
        out.extend(range(start, end - 1, -1))
        idx += 2
        if not idx <= L:
            break
        start, end = at(idx), at(idx + 1)
    return out


def listed_tiles(lists: torch.Tensor) -> int:
    """Sum over rows of sum over ranges (start-end+1) — the corrected statistic (SURVEY B-3)."""
    flat = lists.reshape(-1, lists.shape[-1]).tolist()
    return sum(len(walk_tiles(r)) for r in flat)


def simulate_writer(read_row: Sequence[int], flags: Sequence[bool],
                    must_do_row: Optional[Sequence[int]] = None) -> List[int]:
    """Pure-Python SkipListWriter. `flags[i]` = skip flag of the i-th visited tile (flags[0] is
    ignored: the first tile is recorded with skip=False, mainloop...:1804-1805). Returns the row
    written ([L, ...]) without padding."""
    out = [0]
    state = {"skipping": True}

    def transition(skip, n, md):
        if md is not None and skip:
            if md["end"] > n and md["idx"] <= md["len"]:
                md["idx"] += 2
                md["start"], md["end"] = md["row"][md["idx"]], md["row"][md["idx"] + 1]
            skip = skip and not (n <= md["start"] and n > md["end"])
        if skip != state["skipping"]:
            out.append(n)
            state["skipping"] = skip

    md = None
    if must_do_row is not None:
        r = [int(x) for x in must_do_row] + [0, 0, 0]
        md = {"row": r, "len": r[0], "idx": 1, "start": r[1], "end": r[2]}
    L = int(read_row[0])
    idx = 1
    start, end = int(read_row[1]), int(read_row[2])
    n = start
    pos = 0
    skip = False
    transition(False, n, None)
    n -= 1
    pos += 1
    while True:
        while n >= end:
            skip = bool(flags[pos])
            pos += 1
            transition(skip, n, md)
            n -= 1
        state["skipping"] = True
        if not skip:
            out.append(end)
        idx += 2
        if not idx <= L:
            break
        start, end = int(read_row[idx]), int(read_row[idx + 1])
        n = start
    out[0] = len(out) - 1
    return out


def to_bf16_f32(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def dense_tolerance(out_ref: torch.Tensor, out_pt: torch.Tensor) -> float:
    """Reference tolerance rule (test_flash_attn.py:283,296):
    |out - out_ref| <= 2*|out_pt - out_ref|_max + 2*|out_ref + 0.3 - 0.3 - out_ref|_max."""
    fwd_atol = 2 * (out_ref + 0.3 - 0.3 - out_ref).abs().max().item()
    return 2 * (out_pt.float() - out_ref.float()).abs().max().item() + fwd_atol
