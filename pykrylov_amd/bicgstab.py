"""Bi-CGSTAB behind pykrylov's `BiCGSTAB` class (reference pykrylov/bicgstab/bicgstab.py:9-151)."""
from . import _lib
from .generic import KrylovMethod, solve_guess_matvec_max

__docformat__ = 'restructuredtext'


class BiCGSTAB(KrylovMethod):
    """Bi-Conjugate Gradient Stabilized method for general (nonsymmetric) ``A x = b``.

    Per pass: 2 operator-vector products, 4 dot products + 2 norms, 6 vector updates
    (bicgstab.py:19-20); on the device 2 SpMV kernels with fused dots and 2 streaming kernels
    (``csrc/mk_bicgstab.hip``).
    """

    def __init__(self, op, **kwargs):
        KrylovMethod.__init__(self, op, **kwargs)
        self.name = 'Bi-Conjugate Gradient Stabilized'
        self.acronym = 'Bi-CGSTAB'
        self.prefix = self.acronym + ': '

    def solve(self, rhs, **kwargs):
        """Solve with right-hand side `rhs`.

        :keywords:
            :guess:      initial guess (default 0)
            :matvec_max: max. number of operator-vector products (default 2n)
        """
        solve_guess_matvec_max(self, _lib.MK_BICGSTAB, rhs, kwargs, count_guess_product=True)
