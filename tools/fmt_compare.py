#!/usr/bin/env python3
"""Time y = A x (plain epilogue, mk_spmv) for one matrix in several storage formats.
usage: tools/fmt_compare.py [rows ...]   (random matrices: diagonal + 4 scattered columns per row)"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pykrylov_amd import CsrOperator, _lib  # noqa: E402


def random_csr(n, k=4, seed=1):
    rng = np.random.default_rng(seed)
    cols = np.concatenate([rng.integers(0, n, size=(n, k)), np.arange(n)[:, None]], axis=1)
    cols.sort(axis=1)
    keep = np.ones_like(cols, dtype=bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    indptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int32)
    return indptr, cols[keep].astype(np.int32), rng.standard_normal(int(indptr[-1]))


def main():
    lib = _lib.init()
    sizes = [int(a) for a in sys.argv[1:]] or [100000, 250000, 500000, 1000000]
    for n in sizes:
        ip, ix, dv = random_csr(n)
        x = _lib.DeviceArray.from_numpy(np.random.default_rng(2).standard_normal(n))
        y = _lib.DeviceArray(n)
        alg = 12.0 * len(ix) + 4.0 * (n + 1) + 16.0 * n
        for fmt in (0, 3):
            op = CsrOperator(ip, ix, dv, (n, n))
            _lib.check(lib.mk_csr_set_format(op.handle, fmt))
            f = ctypes.c_int32()
            k = ctypes.c_int32()
            _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(f), None, ctypes.byref(k), None, None))
            for _ in range(5):
                op.spmv_device(x.ptr, y.ptr)
            _lib.check(lib.mk_sync())
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                op.spmv_device(x.ptr, y.ptr)
            _lib.check(lib.mk_sync())
            us = (time.perf_counter() - t0) / reps * 1e6
            print("n=%8d fmt=%d (phases %d): %7.1f us  %.2f TB/s algorithmic" % (n, f.value, k.value, us, alg / us / 1e6))
            op.free()
        x.free()
        y.free()


if __name__ == "__main__":
    main()
