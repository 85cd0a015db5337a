"""`flash_attn.flash_blocksparse_attn_interface` (/root/reference/flash_attn/flash_blocksparse_attn_interface.py:185-200):
static block-sparse attention expressed as skip lists on the gfx950 kernel (thr = -inf: nothing new is dropped)."""
from liteattention_amd.compat import (  # noqa: F401
    flash_blocksparse_attn_qkvpacked_func as flash_blocksparse_attn_func,
    convert_blockmask,
)
