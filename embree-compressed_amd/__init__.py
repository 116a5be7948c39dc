"""embree-compressed on MI355X: host-side Python plumbing over the C-ABI library.

The product is `lib/libembree3.so` (C++ object model + hand-written HIP traversal kernels for gfx950) and the
embree3 headers under `include/`.  This package only offers a ctypes binding (`rtc`) used by the tests and
by bench.py.  The directory name contains a hyphen, so import it with
`importlib.import_module("embree-compressed_amd")`.
"""
from . import rtc  # noqa: F401

__all__ = ["rtc"]
