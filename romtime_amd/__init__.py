"""romtime_amd -- the POD / (M)DEIM / reduced-operator path of KikeM/romtime on AMD MI355X.

The class surface mirrors ``romtime.rom`` and ``romtime.deim`` (src/romtime/rom/__init__.py:1-12,
src/romtime/deim/__init__.py:1-9); the numerics run in hand-written HIP kernels behind the C ABI
of ``include/romtime_hip.h``.  There is no CPU fallback: without the built library and a GPU the
hot-path calls raise ``RomtimeHipError``.
"""
from ._lib import RomtimeHipError
from .base import Reductor
from .deim import DiscreteEmpiricalInterpolation
from .mdeim import MatrixDiscreteEmpiricalInterpolation
from .nonlinear import MatrixDiscreteEmpiricalInterpolationNonlinear
from .pod import DROP_TOLERANCE, orth
from .rom import RomConstructor, RomConstructorMoving, RomConstructorNonlinear

__all__ = [
    "Reductor",
    "orth",
    "DROP_TOLERANCE",
    "RomConstructor",
    "RomConstructorMoving",
    "RomConstructorNonlinear",
    "DiscreteEmpiricalInterpolation",
    "MatrixDiscreteEmpiricalInterpolation",
    "MatrixDiscreteEmpiricalInterpolationNonlinear",
    "RomtimeHipError",
]
