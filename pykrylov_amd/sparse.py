"""Host-side CSR construction helpers (integer work done once, before the matrix goes to HBM)."""
import numpy as np


def coo_to_csr(rows, cols, vals, shape):
    """Canonical CSR (sorted columns, duplicates summed in input order) from coordinate triples.

    Returns ``(indptr int32, indices int32, data float64)``.
    """
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    m, n = int(shape[0]), int(shape[1])
    if rows.size and (rows.min() < 0 or rows.max() >= m or cols.min() < 0 or cols.max() >= n):
        raise ValueError('coordinate index out of range for shape %s' % (shape,))
    order = np.lexsort((cols, rows))              # stable: equal (row, col) keep input order
    r, c, v = rows[order], cols[order], vals[order]
    if r.size:
        new = np.empty(r.size, dtype=bool)
        new[0] = True
        new[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    else:
        new = np.zeros(0, dtype=bool)
    slot = np.cumsum(new) - 1
    data = np.zeros(int(new.sum()), dtype=np.float64)
    np.add.at(data, slot, v)
    ur, uc = r[new], c[new]
    indptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(ur, minlength=m), out=indptr[1:])
    return indptr.astype(np.int32), uc.astype(np.int32), data


def csr_row_slice(indptr, indices, data, r0, r1):
    """Rows [r0, r1) of a CSR matrix as a new CSR triple (views where possible)."""
    lo, hi = int(indptr[r0]), int(indptr[r1])
    return (indptr[r0:r1 + 1] - indptr[r0]).astype(np.int32), indices[lo:hi], data[lo:hi]
