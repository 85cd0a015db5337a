"""liteattention_amd — MI355X-native QK-Skip attention (drop-in for ``lite_attention``).

Public surface = the reference's (/root/reference/hopper/__init__.py:4-6) plus the functional op.
Importing this package loads the HIP extension; it raises if the extension is missing.
"""
__version__ = "0.1.0"

import sys as _sys

from . import _cabi

# `python -m liteattention_amd.build` must be able to run when the library is missing or stale: it is the one
# entry point that does not load it. Everything else fails loudly here, at import — there is no CPU fallback.
_BUILDING = "liteattention_amd.build" in getattr(_sys, "orig_argv", [])
if not _BUILDING:
    _cabi.load()

if not _BUILDING:
    from .flash_attn_interface import (combine_partials, flash_attn_combine, flash_attn_func, get_tile_sizes,  # noqa: E402
                                       skip_list_stats)
    from .lite_attention import LiteAttention, SeqParallelLiteAttention  # noqa: E402
    from .compat import (blockmask_to_skip_lists, fa2_flash_attn_func, flash_attn_varlen_func,  # noqa: E402
                         flash_blocksparse_attn_func, flash_blocksparse_attn_qkvpacked_func)
    from .calibration import calibrate_threshold  # noqa: E402
    from .parallel import (HeadShardedLiteAttention, RingSeqParallelLiteAttention,  # noqa: E402
                           UlyssesLiteAttention)

__all__ = ["LiteAttention", "SeqParallelLiteAttention", "flash_attn_func", "flash_attn_combine", "combine_partials",
           "get_tile_sizes", "skip_list_stats", "fa2_flash_attn_func", "flash_attn_varlen_func",
           "flash_blocksparse_attn_func", "flash_blocksparse_attn_qkvpacked_func", "blockmask_to_skip_lists", "calibrate_threshold",
           "HeadShardedLiteAttention", "UlyssesLiteAttention", "RingSeqParallelLiteAttention", "__version__"]
