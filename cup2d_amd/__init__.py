"""cup2d_amd -- MI355X (gfx950) backend for CUP2D's per-block stencil hot path.

Host-side mirror (Python, ctypes) of the C-ABI in include/cup2d_hip.h.  All compute runs in
hand-written HIP kernels inside cup2d_amd/libcup2d_hip.so; there is no CPU fallback: importing
works anywhere, creating a context without the library or without a GPU raises.
"""
from .lib import Cup2dError, library_path, load_library  # noqa: F401
from .grid import BlockGrid  # noqa: F401
from .simulation import Simulation  # noqa: F401

__all__ = ["BlockGrid", "Simulation", "Cup2dError", "load_library", "library_path"]
