"""Transpose-free QMR behind pykrylov's `TFQMR` class (reference pykrylov/tfqmr/tfqmr.py:8-159)."""
from . import _lib
from .generic import KrylovMethod, solve_guess_matvec_max

__docformat__ = 'restructuredtext'


class TFQMR(KrylovMethod):
    """Transpose-free Quasi-Minimal Residual method for general (nonsymmetric) ``A x = b``.

    Per pass: 2 operator-vector products, 2 dot products + 2 norms, 10 vector updates
    (tfqmr.py:17-18); on the device 2 SpMV kernels with fused row epilogues and 3 streaming kernels
    (``csrc/mk_tfqmr.hip``).  `residNorm` is the quasi-residual estimate and `converged` tests
    ``residNorm * sqrt(m + 1) < threshold`` as in tfqmr.py:156 (a solve that needs no iteration
    reports m = 0 instead of raising `UnboundLocalError` as the reference does).
    """

    def __init__(self, op, **kwargs):
        KrylovMethod.__init__(self, op, **kwargs)
        self.name = 'Transpose-Free Quasi-Minimum Residual'
        self.acronym = 'TFQMR'
        self.prefix = self.acronym + ': '

    def solve(self, rhs, **kwargs):
        """Solve with right-hand side `rhs`.

        :keywords:
            :guess:      initial guess (default 0)
            :matvec_max: max. number of operator-vector products (default 2n)
        """
        solve_guess_matvec_max(self, _lib.MK_TFQMR, rhs, kwargs, count_guess_product=False)
