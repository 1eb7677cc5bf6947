"""Conjugate Gradient Squared behind pykrylov's `CGS` class (reference pykrylov/cgs/cgs.py:8-123)."""
from . import _lib
from .generic import KrylovMethod, solve_guess_matvec_max

__docformat__ = 'restructuredtext'


class CGS(KrylovMethod):
    """Conjugate Gradient Squared method for general (nonsymmetric) ``A x = b``.

    Per pass: 2 operator-vector products, 2 dot products + 1 norm, 7 vector updates (cgs.py:18);
    on the device 2 SpMV kernels (the second with the residual update and both reductions fused
    into its row epilogue) and 2 streaming kernels (``csrc/mk_cgs.hip``).
    """

    def __init__(self, op, **kwargs):
        KrylovMethod.__init__(self, op, **kwargs)
        self.name = 'Conjugate Gradient Squared'
        self.acronym = 'CGS'
        self.prefix = self.acronym + ': '

    def solve(self, rhs, **kwargs):
        """Solve with right-hand side `rhs`.

        :keywords:
            :guess:      initial guess (default 0)
            :matvec_max: max. number of operator-vector products (default 2n)
        """
        solve_guess_matvec_max(self, _lib.MK_CGS, rhs, kwargs, count_guess_product=False)
