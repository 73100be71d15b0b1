"""Alias of wgsassign_amd.utils (same names as the reference module WGSassign/utils)."""
from wgsassign_amd.utils import *  # noqa: F401,F403
from wgsassign_amd import utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
