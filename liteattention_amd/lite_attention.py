"""LiteAttention for MI355X: the per-layer owner of the ping-pong skip lists.

API-compatible with /root/reference/hopper/lite_attention.py (class, method, argument and attribute
names; defaults; list layout ``_skip_list[2, maxB, H, Qt, Kt+1]``; ``_phase``; ``threshold``), so
``from lite_attention import LiteAttention`` code runs unchanged. The attention itself is
``flash_attn_func`` -> ``torch.ops.lite_attention.fwd`` -> C-ABI ``la_fwd`` -> the gfx950 HIP kernel;
list construction lives in ``skip_lists.py``.

Behavioural differences, all fixes of reference defects:
  B-1  ``enable_skipping=False`` runs the dense kernel (reference: AttributeError, lite_attention.py:262-266)
  B-2  K-tile count from ``key.shape[1]`` (reference: from the query length, :121-122)
  B-3  ``calc_percentage`` is the true listed fraction (reference returns about -1, :61-85)
  B-4  ``must_skip_list`` in README format, caller's list untouched, nothing printed (:126-145)
  B-6  must-do list handed to the kernel as one cached row (reference: 4-D repeat + H2D per call, :239-241)
"""
from __future__ import annotations

import os
from typing import Optional, Tuple, Union

import torch

from . import skip_lists as _sl
from .flash_attn_interface import flash_attn_func, get_tile_sizes

Tensor = torch.Tensor


def _verbose() -> bool:
    return os.getenv("LITE_ATTENTION_VERBOSE", "FALSE") != "FALSE"


class LiteAttention:
    """QK-Skip attention with internally managed read/write skip lists.

    Args mirror the reference (lite_attention.py:36): ``enable_skipping=True``, ``threshold=-10.0``
    (log2 domain, must be negative unless env LITE_ATTENTION_DEBUG is set), ``max_batch_size=4`` (an upper bound: the lists are
    allocated for the batch actually seen and grow up to it, ``_skip_list[2, batch_seen, H, Qt, Kt+1]``).
    One instance per attention layer; not thread-safe (README.md:162-172)."""

    def __init__(self, enable_skipping: bool = True, threshold: float = -10.0, max_batch_size: int = 4):
        self._skip_list: Optional[Tensor] = None   # [2, max_batch, H, Qt, Kt+1] int32
        self._phase = 0                            # index of the buffer the NEXT call reads
        self._shape_key = None                     # what the lists were built for
        self._must_do_rows = {}                    # (tokens, block_n, row width, device) -> device row (LRU, _MUST_DO_ROWS_MAX)
        self._last_percentage = 0.0
        self.enable_skipping = enable_skipping
        self.max_batch_size = max_batch_size
        self.set_threshold(threshold)

    # ---- reference-compatible static helpers -------------------------------------------------
    @staticmethod
    def ceil_div(x, y):
        return _sl.cdiv(x, y)

    @staticmethod
    def get_MN(head_dim, element_size, v_colmajor=False):
        """(kTileM, kTileN) of the kernel, from ``la_get_tile_sizes`` (one table shared with the HIP code;
        the reference hand-copies it, lite_attention.py:87-111 vs tile_size.h:10-62)."""
        # v_colmajor: accepted and ignored. In the reference it only switches the fp8 head_dim-128 tile (224 -> 192 keys,
        # tile_size.h:55) - and only on the Python side: its op passes v_colmajor = false to tile_size_fwd_sm90 and requires
        # v.stride(-1) == 1 (flash_api.cpp:413,728), so a column-major V never reaches its kernel. This build's tiles do not
        # depend on the layout of V.
        return get_tile_sizes(head_dim, element_size)

    @staticmethod
    def calc_percentage(read_list: Tensor) -> float:
        """Fraction of tiles that are listed, i.e. NOT skipped (what lite_attention.py:61-85 intended)."""
        return _sl.listed_fraction(read_list)

    @staticmethod
    def init_skip_list(batch, seq_len, heads, head_dim, v_colmajor, dtype, device, must_skip_list=None,
                       seq_len_k=None) -> Tensor:
        """``[2, batch, heads, q_tiles, k_tiles+1]`` int32, rows ``[2, k_tiles-1, 0, ...]`` (:113-153).
        ``seq_len_k`` (extension) defaults to ``seq_len``."""
        _, bn, qt, kt = _sl.tile_geometry(seq_len, seq_len if seq_len_k is None else seq_len_k, head_dim,
                                          dtype.itemsize)
        row = None if must_skip_list is None else _sl.must_skip_row(must_skip_list, bn, kt)
        return _sl.new_skip_lists(batch, heads, qt, kt, device, row)

    @staticmethod
    def _expand_must_do_list(must_do_list, list_shape, query, value):
        """4-D ``[maxB, H, Qt, Kt+1]`` expansion kept for API parity (:214-242); ``__call__`` does not use it."""
        _, bn = get_tile_sizes(query.shape[-1], query.dtype.itemsize)
        row = _sl.must_do_row(must_do_list, bn, list_shape[3], query.device)
        return row.repeat(*list_shape[:3], 1).contiguous()

    # ---- list management ----------------------------------------------------------------------
    def _init_skip_list(self, query: Tensor, value: Tensor, must_skip_list: list = None, batch: Optional[int] = None) -> Tensor:
        """Both ping-pong buffers for ``batch`` sequences (default: the batch of ``query``). The reference always allocates
        ``max_batch_size`` (lite_attention.py:155-162, SURVEY Appendix B-7: 326 MB per layer at the Wan2.1 shape with the
        default 4, 13 GB over 40 layers); here the lists cover the batch actually seen and grow on demand up to
        ``max_batch_size`` (``_get_read_write_lists``), so a default-constructed object costs what ``max_batch_size=1`` would."""
        return self.init_skip_list(query.shape[0] if batch is None else batch, query.shape[1], query.shape[2], query.shape[3],
                                   False, query.dtype, query.device, must_skip_list, seq_len_k=value.shape[1])

    def _get_read_write_lists(self, query: Tensor, value: Tensor, must_skip_list: list = None
                              ) -> Tuple[Optional[Tensor], Optional[Tensor]]:
        """(read, write) views for this call; (re)builds the lists when any of seq lengths / heads /
        head_dim / dtype / device - or the tile geometry of the kernel that will run - changed (:179-200) and flips the
        ping-pong phase (:203-210). A batch larger than any seen so far (<= max_batch_size) grows the lists: the state of the
        sequences already tracked is kept, the new ones start from "all tiles listed"."""
        if not self.enable_skipping:
            return None, None
        # checked on EVERY call (the reference only checks at (re)init, :158, and would index past the lists)
        assert query.shape[0] <= self.max_batch_size, "batch size must be less than or equal to max_batch_size (modify max_batch_size in LiteAttention constructor)"
        # the tile sizes belong to the key: LA_FWD_KERNEL=v2 changes the q-tile of head_dim 128 from 256 to 128 rows, and lists
        # built for one geometry are mis-shaped for the other
        key = (query.shape[1], value.shape[1], query.shape[2], query.shape[3], query.dtype, query.device,
               get_tile_sizes(query.shape[3], query.dtype.itemsize))
        if self._skip_list is None or key != self._shape_key:
            self._skip_list = self._init_skip_list(query, value, must_skip_list)
            self._shape_key = key
            self._phase = 0
            if _verbose():
                print("[Warning]: reinitialized skip list during the forward pass")
        elif query.shape[0] > self._skip_list.shape[1]:
            # growth replaces the tensor: a HIP graph captured earlier keeps pointers into the OLD lists and would ping-pong on freed
            # memory. It cannot be allowed inside a capture, and graphs captured before it must be re-captured (or avoid it: preallocate()).
            if query.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the skip lists would have to grow (a larger batch than any seen so far) inside a HIP-graph capture: "
                                   "call preallocate(batch) before capturing")
            grown = self._init_skip_list(query, value, must_skip_list)
            grown[:, : self._skip_list.shape[1]] = self._skip_list
            self._skip_list = grown
        rd = self._phase
        self._phase = 1 - rd
        return self._skip_list[rd], self._skip_list[1 - rd]

    def preallocate(self, query: Tensor, value: Tensor, batch: Optional[int] = None, must_skip_list: list = None):
        """Size the lists for ``batch`` sequences (default ``max_batch_size``) of this shape NOW, so that no later call has to grow them:
        what a caller does before capturing calls into a HIP graph (a graph keeps the list pointers of its capture; growth replaces the
        tensor). The state of sequences already tracked is kept; (re)initialises like any shape change otherwise."""
        batch = self.max_batch_size if batch is None else batch
        assert batch <= self.max_batch_size, "batch size must be less than or equal to max_batch_size (modify max_batch_size in LiteAttention constructor)"
        key = (query.shape[1], value.shape[1], query.shape[2], query.shape[3], query.dtype, query.device,
               get_tile_sizes(query.shape[3], query.dtype.itemsize))
        if self._skip_list is None or key != self._shape_key:
            self._skip_list = self._init_skip_list(query, value, must_skip_list, batch=batch)
            self._shape_key, self._phase = key, 0
        elif batch > self._skip_list.shape[1]:
            grown = self._init_skip_list(query, value, must_skip_list, batch=batch)
            grown[:, : self._skip_list.shape[1]] = self._skip_list
            self._skip_list = grown

    _MUST_DO_ROWS_MAX = 8          # distinct must_do_list values whose device rows are kept (least recently used go first)

    def _must_do_device_row(self, must_do_list, query: Tensor, width: int) -> Tensor:
        toks = (0, 0) if must_do_list is None else tuple(must_do_list)   # [0,0] = empty must-do (:267)
        _, bn = get_tile_sizes(query.shape[-1], query.dtype.itemsize)
        # the row depends on nothing but (tokens, tile size, row width, device): it survives list re-initialisation, reset and
        # load_state_dict, so a call captured in a HIP graph never needs the host-to-device copy that builds it
        key = (toks, bn, width, str(query.device))
        row = self._must_do_rows.pop(key, None)
        if row is None:
            row = _sl.must_do_row(toks, bn, width, query.device)
            while len(self._must_do_rows) >= self._MUST_DO_ROWS_MAX:
                self._must_do_rows.pop(next(iter(self._must_do_rows)))
        self._must_do_rows[key] = row                                    # (re)inserted last = most recently used
        return row

    # ---- the attention call -------------------------------------------------------------------
    def __call__(self, query: Tensor, key: Tensor, value: Tensor, scale: Optional[float] = None,
                 return_softmax_lse: bool = False, must_do_list: list = None, must_skip_list: list = None, *,
                 q_descale: Optional[Tensor] = None, k_descale: Optional[Tensor] = None,
                 v_descale: Optional[Tensor] = None) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        """One attention call of a denoising step (:244-291).

        query (B, S, H, D) bf16; key/value (B, Sk, H, D). Returns out (B, S, H, D) — the reference
        docstring's (B, S, H*D) is wrong (Appendix B-5) — or ``(out, lse)`` with lse (B, H, S) fp32.
        ``must_do_list``: token ranges ``[start0, end0, ...]`` (descending) never dropped;
        ``must_skip_list``: token ranges dropped from the start, applied when the lists are (re)built.
        ``q/k/v_descale`` (keyword-only extension; the reference class cannot pass them): fp32
        ``(batch, nheads_k)`` dequantisation scales for e4m3 inputs, as ``flash_attn_func`` takes them."""
        read_list, write_list = self._get_read_write_lists(query, key, must_skip_list)
        must_do = None
        if read_list is not None:
            must_do = self._must_do_device_row(must_do_list, query, read_list.shape[3])
        extra = {}
        if q_descale is not None or k_descale is not None or v_descale is not None:      # only when given: the host-logic
            extra = dict(q_descale=q_descale, k_descale=k_descale, v_descale=v_descale)     # tests record the exact call
        if read_list is None:
            # dense calls (enable_skip_optimization(False): the t2t / t2v / v2t calls of the text + video recipe, README.md:225-246) may be
            # split over the keys when they have fewer items than the device has workgroup slots: num_splits = -1 lets the op decide by
            # the reference's heuristic (flash_api.cpp:437, heuristics.h:25-58; the reference class passes nothing, i.e. 1: its default
            # build has no split kernel, hopper/setup.py:48)
            extra["num_splits"] = -1
        output = flash_attn_func(q=query, k=key, v=value, softmax_scale=scale, attn_read_list=read_list,
                                 attn_must_do_list=must_do, attn_write_list=write_list, thr=self.threshold,
                                 return_softmax_lse=return_softmax_lse, **extra)
        if read_list is not None and _verbose():
            self._last_percentage = self.calc_percentage(read_list[: query.shape[0]])
            print(f"[Info]: Percentage of tiles skipped: {1.0 - self._last_percentage:.2%}")
        return output

    def call_windowed(self, query: Tensor, key: Tensor, value: Tensor, q_windows, window_hook=None,
                      scale: Optional[float] = None, return_softmax_lse: bool = False, must_do_list: list = None,
                      must_skip_list: list = None, q_descale=None, k_descale=None, v_descale=None,
                      static_sched: bool = False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        """``__call__`` as several launches, one per q-tile window ``(first q-tile, count)`` (C-ABI ``q_tile_begin`` /
        ``q_tile_count``; tiles of ``get_MN``). Same lists, same ping-pong step, same results; rows of a window are
        final when its launch completes, and ``window_hook(i, out, row_begin, row_end)`` is called right after window i
        is enqueued — the head-sharded driver starts the xGMI all-gather of those rows there. The reference has one
        launch per call (flash_fwd_launch_template.h:359) and no such entry point."""
        from .flash_attn_interface import mha_fwd
        read_list, write_list = self._get_read_write_lists(query, key, must_skip_list)
        must_do = None
        if read_list is not None:
            must_do = self._must_do_device_row(must_do_list, query, read_list.shape[3])
        q, k, v = [x if x.stride(-1) == 1 else x.contiguous() for x in (query, key, value)]
        out, lse, *_ = mha_fwd(q, k, v, q_descale=q_descale, k_descale=k_descale, v_descale=v_descale,
                               softmax_scale=scale, attn_read_list=read_list, attn_must_do_list=must_do,
                               attn_write_list=write_list, thr=self.threshold, _must_do_is_1d=True,
                               _q_windows=q_windows, _window_hook=window_hook, _static_sched=static_sched)
        return (out, lse) if return_softmax_lse else out

    # ---- state control ------------------------------------------------------------------------
    def reset_skip_state(self):
        """Forget the lists; the next call starts from "all tiles listed" (:293-304)."""
        self._skip_list = None
        self._phase = 0
        self._shape_key = None
        self._last_percentage = 0.0

    def set_threshold(self, threshold: float):
        """Threshold must be negative unless env LITE_ATTENTION_DEBUG != "FALSE" (:306-313)."""
        if threshold >= 0 and os.getenv("LITE_ATTENTION_DEBUG", "FALSE") == "FALSE":
            raise ValueError("threshold must be negative when debug mode is not enabled")
        self.threshold = threshold

    def enable_skip_optimization(self, enable: bool = True):
        self.enable_skipping = enable

    # ---- additions: statistics and checkpoint/resume -------------------------
    def current_read_list(self) -> Optional[Tensor]:
        """The list the next call will read (= what the last call wrote)."""
        return None if self._skip_list is None else self._skip_list[self._phase]

    def get_skip_fraction(self, batch: Optional[int] = None) -> float:
        """Fraction of tiles the next call skips (device reduction; synchronises)."""
        rl = self.current_read_list()
        if rl is None:
            return 0.0
        return 1.0 - self.calc_percentage(rl[: (rl.shape[0] if batch is None else batch)])

    def state_dict(self) -> dict:
        """Everything a run needs to continue bit-identically: both ping-pong lists, the phase, the threshold and what the
        lists were built for (shape key incl. the DEVICE they live on). Tensors are returned on the CPU."""
        key = self._shape_key
        return {
            "skip_list": None if self._skip_list is None else self._skip_list.detach().cpu().clone(),
            "phase": self._phase, "threshold": self.threshold, "enable_skipping": self.enable_skipping,
            "max_batch_size": self.max_batch_size,
            "shape_key": None if key is None else (*key[:4], str(key[4]).replace("torch.", "")),
            "device": None if key is None else str(key[5]),
            "tile_sizes": None if key is None else tuple(key[6]),
        }

    def load_state_dict(self, state: dict, device=None):
        """Restore ``state_dict()``. The lists go back to the device they were saved from (``state["device"]``) unless
        ``device`` names another one; a state that records no device (round-1 checkpoints) needs ``device=``. The next call
        must present the same shapes / dtype / device, otherwise the lists are rebuilt as on any shape change (:179-200)."""
        self.threshold = state["threshold"]
        self.enable_skipping = state["enable_skipping"]
        self.max_batch_size = state["max_batch_size"]
        self.reset_skip_state()
        if state["skip_list"] is None:
            return
        if device is None:
            device = state.get("device")
            if device is None:
                raise ValueError("this state_dict records no device: pass load_state_dict(state, device=...)")
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())      # query.device always carries an index
        sq, sk, h, d, dt = state["shape_key"]
        dtype = getattr(torch, dt)
        tiles = get_tile_sizes(d, dtype.itemsize)
        saved = state.get("tile_sizes")                       # absent in round-2 checkpoints: the list shape decides
        if saved is not None and tuple(saved) != tuple(tiles):
            raise ValueError(f"this state was saved for kernel tiles {tuple(saved)}, the kernel selected now uses {tuple(tiles)} "
                             "(LA_FWD_KERNEL differs?): the lists cannot be reused")
        if tuple(state["skip_list"].shape[3:]) != (_sl.cdiv(sq, tiles[0]), _sl.cdiv(sk, tiles[1]) + 1):
            raise ValueError("the saved lists do not match the tile geometry of the kernel selected now")
        self._skip_list = state["skip_list"].to(dev).contiguous()
        self._phase = state["phase"]
        self._shape_key = (sq, sk, h, d, dtype, dev, tuple(tiles))


class SeqParallelLiteAttention:
    """``num_nodes`` independent skip states, one per K/V split, selected by ``split_idx`` (:322-345).
    The (local Q x split j) tile pattern differs per j, hence one state each. Transport of K/V and the
    LSE merge of the partial results stay with the caller; ``flash_attn_combine`` does the merge."""

    def __init__(self, num_nodes: int, enable_skipping: bool = True, threshold: float = -10.0, max_batch_size: int = 4):
        self.num_nodes = num_nodes
        self.lite_attention = [LiteAttention(enable_skipping, threshold, max_batch_size) for _ in range(num_nodes)]
        self.set_threshold(threshold)
        # e4m3 inputs with return_softmax_lse=True: the caller merges partial results by that LSE (README.md:222-250), so the split
        # runs with the reference's form of P and its row sums (fp32 sums of the un-rounded P, softmax.h:275-296; fp32-exact LSE) even
        # when the process opted into a faster form (LA_FP8_P / fwd_flags), whose LSE carries the 8-bit rounding of P. Set False to keep
        # whatever form is selected.
        self.exact_fp8_lse = True

    def __call__(self, query: Tensor, key: Tensor, value: Tensor, split_idx: int, scale: Optional[float] = None,
                 return_softmax_lse: bool = False, *, q_descale: Optional[Tensor] = None, k_descale: Optional[Tensor] = None,
                 v_descale: Optional[Tensor] = None) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        assert split_idx < self.num_nodes, "split_idx must be less than num_nodes"
        kw = {}
        if q_descale is not None or k_descale is not None or v_descale is not None:       # keyword-only extension, as LiteAttention.__call__
            kw = dict(q_descale=q_descale, k_descale=k_descale, v_descale=v_descale)
        if return_softmax_lse and self.exact_fp8_lse and query.dtype == torch.float8_e4m3fn:
            from .flash_attn_interface import fwd_flags
            from ._cabi import LA_FLAG_FP8_ENCODED_P, LA_FLAG_FP8_MFMA_ROWSUM
            with fwd_flags(0, clear=LA_FLAG_FP8_ENCODED_P | LA_FLAG_FP8_MFMA_ROWSUM):
                return self.lite_attention[split_idx](query, key, value, scale, return_softmax_lse, **kw)
        return self.lite_attention[split_idx](query, key, value, scale, return_softmax_lse, **kw)

    def reset_skip_state(self):
        for la in self.lite_attention:
            la.reset_skip_state()

    def set_threshold(self, threshold: float):
        for la in self.lite_attention:
            la.set_threshold(threshold)

    def enable_skip_optimization(self, enable: bool = True):
        for la in self.lite_attention:
            la.enable_skip_optimization(enable)
