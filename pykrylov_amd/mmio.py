"""MatrixMarket coordinate reader -> device CSR operator.

Replaces the Pysparse loader the reference's examples rely on
(reference examples/demo_common.py:12-16: ``spmatrix.ll_mat_from_mtx`` + ``PysparseLinearOperator``).
"""
import numpy as np

from .sparse import coo_to_csr


def read_mtx_coo(path):
    """Parse a MatrixMarket *coordinate* file; returns ``(rows, cols, vals, shape, symmetric)`` (0-based triples)
    with symmetric / skew-symmetric storage expanded to the full matrix."""
    with open(path, 'r') as fh:
        banner = fh.readline().split()
        if len(banner) < 5 or banner[0].lower() != '%%matrixmarket' or banner[1].lower() != 'matrix':
            raise ValueError('%s: not a MatrixMarket matrix file' % path)
        layout, field, symm = (t.lower() for t in banner[2:5])
        if layout != 'coordinate':
            raise ValueError('%s: only coordinate format is supported' % path)
        if field not in ('real', 'integer', 'pattern'):
            raise ValueError('%s: field %r is not supported (real data only)' % (path, field))
        if symm not in ('general', 'symmetric', 'skew-symmetric'):
            raise ValueError('%s: symmetry %r is not supported' % (path, symm))
        line = fh.readline()
        while line and (line.startswith('%') or not line.strip()):
            line = fh.readline()
        m, n, nz = (int(t) for t in line.split())
        want = 2 if field == 'pattern' else 3
        flat = np.array(fh.read().split(), dtype=np.float64)
    if flat.size != nz * want:
        raise ValueError('%s: expected %d entries, found %d numbers' % (path, nz, flat.size))
    body = flat.reshape(nz, want)
    rows = body[:, 0].astype(np.int64) - 1
    cols = body[:, 1].astype(np.int64) - 1
    vals = np.ones(nz) if field == 'pattern' else body[:, 2].copy()
    if symm != 'general':
        off = rows != cols
        sign = -1.0 if symm == 'skew-symmetric' else 1.0
        rows, cols, vals = (np.concatenate([rows, cols[off]]), np.concatenate([cols, rows[off]]),
                            np.concatenate([vals, sign * vals[off]]))
    return rows, cols, vals, (m, n), symm == 'symmetric'


def read_mtx(path):
    """Parse a MatrixMarket *coordinate* file on the host; returns ``(indptr, indices, data, shape, symmetric)``
    with symmetric / skew-symmetric storage expanded to the full matrix."""
    rows, cols, vals, shape, symmetric = read_mtx_coo(path)
    indptr, indices, data = coo_to_csr(rows, cols, vals, shape)
    return indptr, indices, data, shape, symmetric


def csr_operator_from_mtx(path):
    """Device-resident operator for the matrix stored in `path`; the coordinate triples are sorted and
    assembled into CSR on the GPU (``mk_csr_from_coo``), bit-identical to the host construction."""
    from .linop import CsrOperator
    rows, cols, vals, shape, symmetric = read_mtx_coo(path)
    return CsrOperator.from_coo(rows, cols, vals, shape, symmetric=symmetric)
