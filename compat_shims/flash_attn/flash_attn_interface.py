"""FlashAttention-2 signatures (/root/reference/flash_attn/flash_attn_interface.py:1014-1450) on the gfx950 kernel."""
from liteattention_amd.compat import (  # noqa: F401
    fa2_flash_attn_func as flash_attn_func,
    fa2_flash_attn_kvpacked_func as flash_attn_kvpacked_func,
    fa2_flash_attn_qkvpacked_func as flash_attn_qkvpacked_func,
    fa2_flash_attn_varlen_func as flash_attn_varlen_func,
    fa2_flash_attn_varlen_kvpacked_func as flash_attn_varlen_kvpacked_func,
    fa2_flash_attn_varlen_qkvpacked_func as flash_attn_varlen_qkvpacked_func,
)


def flash_attn_with_kvcache(*args, **kwargs):
    raise NotImplementedError("flash_attn_with_kvcache (paged / appended KV cache) is outside the QK-Skip hot path of this build")
