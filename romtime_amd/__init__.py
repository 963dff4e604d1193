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



def shutdown():
    """Release what the package holds on the device, in a fixed order, while the interpreter is fully alive: the calling
    thread's sequence runners (worker threads joined), the process-wide CU-masked streams, every context.  Optional - the
    same happens piecemeal at interpreter exit - but a long-lived host (a test session, a service) that is done with the
    package can call it to give the memory back; the package works again afterwards (new contexts and streams are made on
    demand).  Pipelines and runners created before the call must not be used after it."""
    import torch

    from . import _lib, pipeline, walks

    walks.close_runners()
    pipeline.shutdown()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _lib.Context.destroy_all()


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
    "shutdown",
]
