"""pykrylov_amd -- an MI355X (gfx950) Krylov inner-loop engine behind pykrylov's API.

`LinearOperator` / `KrylovMethod` protocol of PythonOptimizers/pykrylov, with the CSR SpMV
and the per-iteration dot / norm / axpy work of the solver loops running as hand-written HIP
kernels in ``libmikrylov.so`` (C ABI: ``include/mikrylov.h``, bound with ctypes).
There is no CPU fallback: without the shared object or without a GPU the solvers raise.
"""
__version__ = '0.1.0'

from .linop import (BaseLinearOperator, LinearOperator, IdentityOperator, DiagonalOperator,     # noqa: F401
                    ZeroOperator, ReducedLinearOperator, SymmetricallyReducedLinearOperator,
                    CoordLinearOperator, CsrOperator, ShapeError, linop_from_ndarray)
from .generic import KrylovMethod                                                               # noqa: F401
from .cg import CG                                                                              # noqa: F401
from .bicgstab import BiCGSTAB                                                                  # noqa: F401
from .cgs import CGS                                                                            # noqa: F401
from .tfqmr import TFQMR                                                                        # noqa: F401
from .minres import Minres                                                                      # noqa: F401
from .symmlq import Symmlq                                                                      # noqa: F401
from . import lls                                                                               # noqa: F401
from . import blkop                                                                             # noqa: F401
