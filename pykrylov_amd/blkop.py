"""Block operators: a grid (or a diagonal) of linear operators seen as one operator.

Host-side composition, as in the reference (pykrylov/linop/blkop.py:8-276): the blocks may be any operators of the
protocol -- device matrices included, whose products then run on the GPU one block at a time -- and the result is an
ordinary `LinearOperator`; the device solvers call it back at their product sites (HostOperatorShell, linop.py).
Every block row is accumulated in the order the reference uses, ``y_i = ((0 + B_i0 x_0) + B_i1 x_1) + ...``
(blkop.py:86-96), so results carry the same roundings.
"""
import itertools

import numpy as np

from .linop import BaseLinearOperator, LinearOperator, ShapeError, null_log

__docformat__ = 'restructuredtext'


def _offsets(sizes):
    return np.concatenate([[0], np.cumsum(sizes)]).astype(int)


def _grid_sizes(grid):
    """(heights of the block rows, widths of the block columns) of a rectangular grid of operators."""
    widths = [blk.shape[1] for blk in grid[0]]
    heights = []
    for row in grid:
        if len(row) != len(widths) or [blk.shape[1] for blk in row] != widths:
            raise ShapeError('Inconsistent block shapes')
        tall = set(blk.shape[0] for blk in row)
        if len(tall) != 1:
            raise ShapeError('Inconsistent block shapes')
        heights.append(tall.pop())
    return heights, widths


def _transposed(grid, attr):
    return [[getattr(grid[i][j], attr) for i in range(len(grid))] for j in range(len(grid[0]))]


class BlockLinearOperator(LinearOperator):
    """Operator given block row by block row, ``[[A, B], [C, D]]`` (one row: ``[[A, B]]``).  With ``symmetric=True``
    (or ``hermitian=True``) only the upper triangle is listed, ``[[A, B, C], [D, E], [F]]``; the diagonal blocks
    must then be symmetric (hermitian) and the lower triangle is filled with transposes (adjoints) of the given
    blocks -- references, not copies (blkop.py:8-44)."""

    def __init__(self, blocks, symmetric=False, hermitian=False, **kwargs):
        grid = [list(row) for row in blocks]
        if symmetric or hermitian:
            if len(grid) != len(grid[0]):
                raise ShapeError('Inconsistent shape.')
            for row in grid:
                if symmetric and not row[0].symmetric:
                    raise ValueError('Blocks on diagonal must be symmetric.')
                if hermitian and not row[0].hermitian:
                    raise ValueError('Blocks on diagonal must be hermitian.')
            # row i lists blocks (i, i) ... (i, n-1); block (i, j) with j < i mirrors block (j, i) = upper[j][i - j]
            upper = grid
            mirror = 'T' if symmetric else 'H'
            grid = [[getattr(upper[j][i - j], mirror) for j in range(i)] + list(upper[i]) for i in range(len(upper))]
        self._blocks = grid
        log = kwargs.get('logger', null_log)
        log.debug('Building new BlockLinearOperator')
        heights, widths = _grid_sizes(grid)
        self._blocksT = _transposed(grid, 'T')
        self._blocksH = _transposed(grid, 'H')
        dtype = np.result_type(*[blk.dtype for blk in itertools.chain(*grid)])
        super(BlockLinearOperator, self).__init__(
            sum(widths), sum(heights), symmetric=symmetric, hermitian=hermitian, dtype=dtype,
            matvec=lambda x: self._apply(x, '_blocks'),
            matvec_transp=lambda x: self._apply(x, '_blocksT'),
            matvec_adj=lambda x: self._apply(x, '_blocksH'))
        if self.T is not None and self.T is not self:
            self.T._blocks = self._blocksT
        if self.H is not None and self.H is not self:
            self.H._blocks = self._blocksH

    def _apply(self, x, which):
        grid = getattr(self, which)
        heights, widths = _grid_sizes(grid)
        x = np.asarray(x)
        if len(x) != sum(widths):
            raise ShapeError('Multiplying with vector of wrong shape.')
        self.logger.debug('Multiplying with a vector of size %d' % len(x))
        y = np.zeros(sum(heights), dtype=np.result_type(self.dtype, x.dtype))
        r, c = _offsets(heights), _offsets(widths)
        for i, row in enumerate(grid):
            out = y[r[i]:r[i + 1]]
            for j, blk in enumerate(row):
                out[:] += blk * x[c[j]:c[j + 1]]              # (blkop.py:94: in this order, one block at a time)
        return y

    def _device_view(self):
        """The block operator as ONE device operator (linop._BlockCsrOperator) when every block is a device matrix or
        a plain diagonal / identity operator; None otherwise (the device solvers then call the host composition back).
        Rebuilt when blocks were replaced."""
        from .linop import _BlockCsrOperator, device_block
        grid = self._blocks
        key = tuple(id(b) for row in grid for b in row)
        cached = self.__dict__.get('_dev_view')
        if cached is not None and cached[0] == key and cached[1].handle:
            return cached[1]
        dev = [[device_block(b) for b in row] for row in grid]
        if any(d is None for row in dev for d in row):
            return None
        heights, widths = _grid_sizes(grid)
        view = _BlockCsrOperator.build(self, dev, heights, widths)
        if view is not None:
            self.__dict__['_dev_view'] = (key, view)
        return view

    @property
    def blocks(self):
        "The list of lists of blocks."
        return self._blocks

    def _as_array(self):
        arr = np.empty((len(self._blocks), len(self._blocks[0])), dtype=object)
        for i, row in enumerate(self._blocks):
            for j, blk in enumerate(row):
                arr[i, j] = blk
        return arr

    def __getitem__(self, indices):
        sel = self._as_array()[indices]
        if isinstance(sel, BaseLinearOperator):
            return sel
        if sel.ndim == 1:
            # one index was an integer: a column of blocks if it was the second one, else a row (np.matrix keeps
            # both two-dimensional, blkop.py:123-129)
            column = isinstance(indices, tuple) and len(indices) == 2 and isinstance(indices[1], (int, np.integer))
            sel = sel.reshape(-1, 1) if column else sel.reshape(1, -1)
        return BlockLinearOperator(sel.tolist(), symmetric=False, hermitian=False)

    def __setitem__(self, indices, val):
        arr = self._as_array()
        arr[indices] = val
        grid = arr.tolist()
        if self.symmetric or self.hermitian:
            mirror = 'T' if self.symmetric else 'H'
            for i in range(1, len(grid)):
                for j in range(i):
                    grid[i][j] = getattr(grid[j][i], mirror)
        _grid_sizes(grid)
        self._blocks = grid
        self._blocksT = _transposed(grid, 'T')
        self._blocksH = _transposed(grid, 'H')
        if self.T is not None and self.T is not self:
            self.T._blocks = self._blocksT
        if self.H is not None and self.H is not self and self.H is not self.T:
            self.H._blocks = self._blocksH

    def __contains__(self, op):
        return any(op is blk or op == blk for blk in itertools.chain(*self._blocks))

    def __iter__(self):
        for row in self._blocks:
            yield row


class BlockDiagonalLinearOperator(LinearOperator):
    """``diag(A, B, C)`` from the list ``[A, B, C]`` (blkop.py:154-257); symmetric when every block is."""

    def __init__(self, blocks, **kwargs):
        blocks = list(blocks)
        self._blocks = blocks
        log = kwargs.get('logger', null_log)
        log.debug('Building new BlockDiagonalLinearOperator')
        symmetric = all(blk.symmetric for blk in blocks)
        hermitian = all(blk.hermitian for blk in blocks)
        self._blocksT = [blk.T for blk in blocks]
        self._blocksH = [blk.H for blk in blocks]
        dtype = np.result_type(*[blk.dtype for blk in blocks])
        super(BlockDiagonalLinearOperator, self).__init__(
            sum(blk.shape[1] for blk in blocks), sum(blk.shape[0] for blk in blocks),
            symmetric=symmetric, hermitian=hermitian, dtype=dtype,
            matvec=lambda x: self._apply(x, '_blocks'),
            matvec_transp=lambda x: self._apply(x, '_blocksT'),
            matvec_adj=lambda x: self._apply(x, '_blocksH'))
        if self.T is not None and self.T is not self:
            self.T._blocks = self._blocksT
        if self.H is not None and self.H is not self:
            self.H._blocks = self._blocksH

    def _apply(self, x, which):
        blks = getattr(self, which)
        x = np.asarray(x)
        r = _offsets([blk.shape[0] for blk in blks])
        c = _offsets([blk.shape[1] for blk in blks])
        if len(x) != c[-1]:
            raise ShapeError('Multiplying with vector of wrong shape.')
        self.logger.debug('Multiplying with a vector of size %d' % len(x))
        y = np.empty(r[-1], dtype=np.result_type(self.dtype, x.dtype))
        for k, blk in enumerate(blks):
            y[r[k]:r[k + 1]] = blk * x[c[k]:c[k + 1]]
        return y

    def _device_view(self):
        "As `BlockLinearOperator._device_view`: diag(A, B, ...) of device matrices as one device operator, or None."
        from .linop import _BlockCsrOperator, device_block
        blks = self._blocks
        key = tuple(id(b) for b in blks)
        cached = self.__dict__.get('_dev_view')
        if cached is not None and cached[0] == key and cached[1].handle:
            return cached[1]
        dev = [device_block(b) for b in blks]
        if any(d is None for d in dev):
            return None
        grid = [[dev[i] if i == j else None for j in range(len(dev))] for i in range(len(dev))]
        view = _BlockCsrOperator.build(self, grid, [b.shape[0] for b in blks], [b.shape[1] for b in blks])
        if view is not None:
            self.__dict__['_dev_view'] = (key, view)
        return view

    @property
    def blocks(self):
        "The list of diagonal blocks."
        return self._blocks

    def __getitem__(self, idx):
        sel = self._blocks[idx]
        if isinstance(idx, slice):
            return BlockDiagonalLinearOperator(sel)
        return sel

    def __setitem__(self, idx, ops):
        new = ops if isinstance(ops, (list, tuple)) else [ops]
        for op in new:
            if not isinstance(op, BaseLinearOperator):
                raise ValueError('Block operators can only contain linear operators')
        self._blocks[idx] = ops
        self._blocksT = [blk.T for blk in self._blocks]
        self._blocksH = [blk.H for blk in self._blocks]
        if self.T is not None and self.T is not self:
            self.T._blocks = self._blocksT
        if self.H is not None and self.H is not self and self.H is not self.T:
            self.H._blocks = self._blocksH


class BlockPreconditioner(BlockLinearOperator):
    "A `BlockLinearOperator` whose ``solve`` applies it (blkop.py:259-266)."

    def solve(self, x):
        return self.__call__(x)


class BlockDiagonalPreconditioner(BlockDiagonalLinearOperator):
    "A `BlockDiagonalLinearOperator` whose ``solve`` applies it (blkop.py:269-276)."

    def solve(self, x):
        return self.__call__(x)
