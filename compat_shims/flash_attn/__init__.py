"""`flash_attn` import name on MI355X: the FlashAttention-2 functional surface
(/root/reference/flash_attn/__init__.py:3-11) forwarded to the gfx950 QK-Skip forward kernel in dense mode.
Opt-in: only importable when `compat_shims/` is on sys.path (see compat_shims/README.md)."""
__version__ = "2.8.3+liteattention_amd"

from flash_attn.flash_attn_interface import (  # noqa: F401
    flash_attn_func,
    flash_attn_kvpacked_func,
    flash_attn_qkvpacked_func,
    flash_attn_varlen_func,
    flash_attn_varlen_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
    flash_attn_with_kvcache,
)
