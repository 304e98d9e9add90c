"""t8gpu_amd -- MI355X-native backend for the t8gpu finite-volume hot path.

Only what the path needs: csrc/ (HIP kernels + C-ABI + host-side mesh/plan code)
and thin ctypes mirrors used by tests/ and bench.py. The product interface is the
C-ABI in include/t8gpu_hip.h and the C++ headers in include/t8gpu/.
"""
from . import build  # noqa: F401

__all__ = ["build", "synth", "hip", "solver"]
