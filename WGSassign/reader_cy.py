"""Alias of wgsassign_amd.reader_cy (same names as the reference module WGSassign/reader_cy)."""
from wgsassign_amd.reader_cy import *  # noqa: F401,F403
from wgsassign_amd import reader_cy as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
