"""Least-squares solvers behind pykrylov's `lls` classes (reference pykrylov/lls/*.py).

`LSQRFramework`, `LSMRFramework`, `CRAIGFramework`, `CRAIGMRFramework` keep the reference's `solve`
signatures and result attributes; the Golub-Kahan bidiagonalisation (one product with A and one with
A' per iteration) and every scalar recurrence run on the GPU (``csrc/mk_lls.hip``).  With a
:class:`pykrylov_amd.linop.CsrOperator` both products run on the device (the transpose is built there on first
use); any other operator with ``A * v`` and ``A.T * u`` is called back on the host at each product site
(:class:`pykrylov_amd.linop.HostOperatorShell`).  Preconditioners `M`, `N` run on the device when they expose a
diagonal (``DiagonalOperator``, linop.py:473-516; ``mk_solver_set_lls_precon``); any other callable `M` / `N` is
called back on the host at its `u = M(Mu)` / `v = N(Nv)` sites (``mk_solver_set_lls_precon_callback``).  On several GPUs
the operator is split into row blocks with a replicated column space (:func:`pykrylov_amd.dist.partition_row_blocks`).
"""
from .solvers import LSQRFramework, LSMRFramework, CRAIGFramework, CRAIGMRFramework   # noqa: F401
