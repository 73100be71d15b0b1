"""Alias of wgsassign_amd.glassy (same names as the reference module WGSassign/glassy)."""
from wgsassign_amd.glassy import *  # noqa: F401,F403
from wgsassign_amd import glassy as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
