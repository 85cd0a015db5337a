"""FlashAttention-3 import name (`import flash_attn_interface`, /root/reference/hopper/_internal/flash_attn_interface.py:
547-686) on the gfx950 kernel — with the LiteAttention extensions (attn_read_list / attn_write_list / thr) of the reference."""
from liteattention_amd.flash_attn_interface import (  # noqa: F401
    FlashAttnFunc,
    _flash_attn_forward,
    flash_attn_combine,
    flash_attn_func,
)
from liteattention_amd.compat import (  # noqa: F401
    fa3_flash_attn_qkvpacked_func as flash_attn_qkvpacked_func,
    fa3_flash_attn_varlen_func as flash_attn_varlen_func,
)


def flash_attn_with_kvcache(*args, **kwargs):
    raise NotImplementedError("flash_attn_with_kvcache (paged / appended KV cache) is outside the QK-Skip hot path of this build")


def get_scheduler_metadata(*args, **kwargs):
    raise NotImplementedError("get_scheduler_metadata: the gfx950 kernel schedules itself (ticket queues); nothing to precompute")
