"""tracy_amd -- MI355X-native (gfx950) implementation of tracy's trace-to-reference alignment and
allele-deconvolution hot path.  The product is the C-ABI library tracy_amd/lib/libtracy_hip.so
(include/tracy_hip.h); this package holds its sources (csrc/), the C++ host mirror of the reference's
interface (host/) and a thin ctypes binding used by tests and bench.py.  There is no CPU fallback."""
from . import capi, hostlib  # noqa: F401
from .capi import Context, Group, TracyHipError, library_path  # noqa: F401

__version__ = "0.1.0"
