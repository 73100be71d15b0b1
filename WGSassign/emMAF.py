"""Alias of wgsassign_amd.emMAF (same names as the reference module WGSassign/emMAF)."""
from wgsassign_amd.emMAF import *  # noqa: F401,F403
from wgsassign_amd import emMAF as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
