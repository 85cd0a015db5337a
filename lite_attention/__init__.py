"""Drop-in import name: ``from lite_attention import LiteAttention`` (reference package name,
/root/reference/hopper/setup.py:36,650-651) resolves to the MI355X build."""
from liteattention_amd import *  # noqa: F401,F403
from liteattention_amd import __version__, __all__  # noqa: F401
