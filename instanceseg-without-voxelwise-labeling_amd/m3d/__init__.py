"""m3d — host-side Python over libm3d.so (hand-written HIP for gfx950, C ABI in include/m3d.h).

PyTorch is used for device memory, streams and torch.distributed only.  There is NO CPU fallback: importing
an op without the built library, or calling one without a GPU tensor, raises.
"""
from . import _lib  # noqa: F401
from .ops import *  # noqa: F401,F403
