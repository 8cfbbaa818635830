"""qmps_amd - MI355X-native implementation of qmps's classical inner loop.

Host side: plain Python + numpy calling hand-written gfx950 HIP kernels through the ctypes
C-ABI in include/qmps_hip.h (no PyTorch, no Triton on the product path).  The sub-modules
`tools`, `represent`, `ground_state`, `rotosolve`, `time_evolve_tools` mirror the reference's
API surface (fergusfinn/qmps) so existing driver scripts keep working; `engine.EnergyEngine`
is the batched entry point.
"""
from . import _lib  # noqa: F401
from .engine import EnergyEngine  # noqa: F401

__all__ = ['EnergyEngine']
