"""Alias of wgsassign_amd.fisher (same names as the reference module WGSassign/fisher)."""
from wgsassign_amd.fisher import *  # noqa: F401,F403
from wgsassign_amd import fisher as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in __all__})
