"""Least-squares solvers behind pykrylov's `lls` classes (reference pykrylov/lls/*.py).

`LSQRFramework`, `LSMRFramework`, `CRAIGFramework`, `CRAIGMRFramework` keep the reference's `solve`
signatures and result attributes; the Golub-Kahan bidiagonalisation (one product with A and one with
A' per iteration) and every scalar recurrence run on the GPU (``csrc/mk_lls.hip``).  The operator must
be a :class:`pykrylov_amd.linop.CsrOperator`; its transpose is built on the device on first use.
Preconditioners `M`, `N` are not available on the device path yet.
"""
from .solvers import LSQRFramework, LSMRFramework, CRAIGFramework, CRAIGMRFramework   # noqa: F401
