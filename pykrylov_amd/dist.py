"""Row-partitioned multi-GPU execution: one process per GPU, RCCL over xGMI.

The reference is single-process; this module is new design.  The matrix is
split into contiguous row blocks, every solver vector likewise.  Before each SpMV a rank needs
the entries of the input vector owned by other ranks that its rows reference:

* ``halo``      -- neighbour send/recv of exactly those entries (2 planes of 512^2 doubles for
                   the slab-partitioned 512^3 Poisson problem); the path that fits the xGMI budget;
* ``allgather`` -- RCCL all-gather of the whole iterate, the general path north_star names.

Dot products are all-reduced as vectors of per-workgroup partial sums (``csrc/mk_solver.hip``).
The integer planning below is plain NumPy so that it can be tested on CPU (world_size 2, gloo);
the exchange itself runs in ``libmikrylov.so`` (``mk_exchange``), enqueued on the solver's stream.
"""
import ctypes
import os

import numpy as np

from . import _lib


# ======================================================================================
# pure planning (no GPU, no communication)
# ======================================================================================
def row_ranges(n, nranks, align=1):
    """Contiguous, balanced row blocks; block starts are multiples of `align` (e.g. a grid plane)."""
    units = n // align
    assert units * align == n, "n must be a multiple of align"
    base, extra = divmod(units, nranks)
    starts = [0]
    for r in range(nranks):
        starts.append(starts[-1] + (base + (1 if r < extra else 0)) * align)
    return [(starts[r], starts[r + 1]) for r in range(nranks)]


def equal_ranges(n, nranks):
    """Blocks of ceil(n / nranks) rows (the layout an all-gather needs); trailing blocks may be short."""
    cnt = -(-n // nranks)
    if cnt * (nranks - 1) >= n:
        raise NotImplementedError("all-gather exchange: %d rows leave a rank without rows on %d ranks; "
                                  "use fewer ranks or mode='halo'" % (n, nranks))
    return [(min(n, r * cnt), min(n, (r + 1) * cnt)) for r in range(nranks)], cnt


def owner_of(cols, ranges):
    starts = np.array([r[0] for r in ranges] + [ranges[-1][1]], dtype=np.int64)
    return np.searchsorted(starts, cols, side="right") - 1


def needed_columns(indices, c0, c1):
    """Sorted unique global columns outside the owned range [c0, c1)."""
    idx = np.asarray(indices, dtype=np.int64)
    off = idx[(idx < c0) | (idx >= c1)]
    return np.unique(off)


def localize_columns(indices, c0, c1, halo_cols):
    """Global -> local column ids: owned columns first, then `halo_cols` in ascending order."""
    idx = np.asarray(indices, dtype=np.int64)
    out = idx - c0
    off = (idx < c0) | (idx >= c1)
    out[off] = (c1 - c0) + np.searchsorted(halo_cols, idx[off])
    return out.astype(np.int32)


def halo_plan(rank, ranges, needs_by_rank):
    """Exchange plan of `rank` from every rank's sorted needed-column list.

    Returns ``(send_count[nranks], recv_count[nranks], send_idx)`` where `send_idx` holds local
    row indices grouped by destination rank and the halo region of this rank is the
    concatenation, in rank order, of what it receives (= its own needs in ascending order).
    """
    nranks = len(ranges)
    c0, c1 = ranges[rank]
    recv_count = np.zeros(nranks, dtype=np.int64)
    mine = np.asarray(needs_by_rank[rank], dtype=np.int64)
    if mine.size:
        own = owner_of(mine, ranges)
        assert not np.any(own == rank)
        recv_count = np.bincount(own, minlength=nranks).astype(np.int64)
    send_count = np.zeros(nranks, dtype=np.int64)
    send_idx = []
    for r in range(nranks):
        if r == rank:
            continue
        want = np.asarray(needs_by_rank[r], dtype=np.int64)
        sel = want[(want >= c0) & (want < c1)]
        send_count[r] = sel.size
        send_idx.append((sel - c0).astype(np.int32))
    send_idx = np.concatenate(send_idx) if send_idx else np.zeros(0, dtype=np.int32)
    return send_count, recv_count, send_idx


def banded_needs(c0, c1, halo_lo, halo_hi):
    """Needed columns of a banded row block: the two contiguous windows next to the owned range."""
    return np.concatenate([np.arange(c0 - halo_lo, c0, dtype=np.int64),
                           np.arange(c1, c1 + halo_hi, dtype=np.int64)])


# ======================================================================================
# process group / communicator bootstrap
# ======================================================================================
class World(object):
    """Rank bookkeeping + a tiny object all-gather used only while planning.

    `torch.distributed` (already initialised by the launcher contract of bench.py, backend nccl
    on GPUs, gloo in CPU tests) carries the planning metadata and the 128-byte RCCL id; the
    per-iteration collectives never go through torch."""

    def __init__(self, rank=0, nranks=1, torch_dist=None):
        self.rank, self.nranks, self.td = rank, nranks, torch_dist

    @classmethod
    def from_env(cls):
        nranks = int(os.environ.get("WORLD_SIZE", "1"))
        if nranks == 1:
            return cls()
        import torch.distributed as td
        assert td.is_initialized(), "initialise torch.distributed before pykrylov_amd.dist.World.from_env()"
        return cls(td.get_rank(), td.get_world_size(), td)

    def allgather_object(self, obj):
        if self.nranks == 1:
            return [obj]
        out = [None] * self.nranks
        self.td.all_gather_object(out, obj)
        return out

    def _init_host_comm(self, lib):
        self._host_cbs = _host_comm_callbacks(self)          # keep the ctypes thunks alive
        _lib.check(lib.mk_comm_init_host(self.nranks, self.rank, *self._host_cbs))

    def init_device_comm(self, transport="rccl"):
        """Create the communicator inside libmikrylov (one per process).

        `transport="rccl"`: RCCL over xGMI, one process per GPU (the production path).
        `transport="host"`: the same collectives staged through host memory and carried by `torch.distributed`
        (any backend that moves CPU tensors, i.e. gloo) -- slow, but it lets several ranks share one GPU, which
        RCCL refuses; used to test the multi-rank device path on a single-GPU box."""
        lib = _lib.init()
        if transport == "host":
            return self._init_host_comm(lib)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.mk_comm_unique_id(buf))
        uid = self.allgather_object(bytes(buf.raw))[0]
        _lib.check(lib.mk_comm_init(self.nranks, self.rank, ctypes.c_char_p(uid)))


def _host_comm_callbacks(world):
    """ctypes callbacks for mk_comm_init_host built on torch.distributed (CPU tensors)."""
    import torch
    td, nranks, rank = world.td, world.nranks, world.rank

    def view(ptr, count):
        if count == 0:
            return torch.zeros(0, dtype=torch.float64)
        arr = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,))
        return torch.from_numpy(arr)

    def allreduce(buf, count):
        try:
            if nranks > 1:
                td.all_reduce(view(buf, count))
            return 0
        except Exception:                                    # pragma: no cover - reported through mk_last_error
            import traceback
            traceback.print_exc()
            return 1

    def exchange(send, send_count, send_off, recv, recv_count, recv_off):
        try:
            stotal = sum(send_count[r] for r in range(nranks))
            rtotal = sum(recv_count[r] for r in range(nranks))
            s, rv = view(send, stotal), view(recv, rtotal)
            reqs = []
            for r in range(nranks):
                if r != rank and recv_count[r] > 0:
                    reqs.append(td.irecv(rv[recv_off[r]:recv_off[r] + recv_count[r]], src=r))
            for r in range(nranks):
                if r != rank and send_count[r] > 0:
                    reqs.append(td.isend(s[send_off[r]:send_off[r] + send_count[r]].clone(), dst=r))
            for q in reqs:
                q.wait()
            return 0
        except Exception:                                    # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    def allgather(send, count, recv):
        try:
            out = view(recv, count * nranks)
            if nranks > 1:
                td.all_gather([out[r * count:(r + 1) * count] for r in range(nranks)], view(send, count).clone())
            else:
                out.copy_(view(send, count))
            return 0
        except Exception:                                    # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    return (_lib.HOST_ALLREDUCE_FN(allreduce), _lib.HOST_EXCHANGE_FN(exchange), _lib.HOST_ALLGATHER_FN(allgather))


def attach_exchange(op, mode, n_local, n_halo, send_count=None, recv_count=None, send_idx=None):
    lib = _lib.init()
    sc = None if send_count is None else np.ascontiguousarray(send_count, dtype=np.int64)
    rc = None if recv_count is None else np.ascontiguousarray(recv_count, dtype=np.int64)
    si = None if send_idx is None else np.ascontiguousarray(send_idx, dtype=np.int32)
    _lib.check(lib.mk_csr_set_exchange(op.handle, mode, n_local, n_halo,
                                       None if sc is None else sc.ctypes.data,
                                       None if rc is None else rc.ctypes.data,
                                       None if si is None or si.size == 0 else si.ctypes.data))
    op.local_size = int(n_local)
    op.halo_size = int(n_halo)
    op.exchange_mode = int(mode)                              # 0 halo ([own | received] columns), 1 all-gather
    return op


def plan_host_csr(world, indptr, indices, data, n, mode="halo"):
    """Pure planning step of :func:`partition_host_csr` (NumPy + one object all-gather; no GPU):
    the local CSR triple with localized columns, the exchange plan and the row ranges."""
    from .sparse import csr_row_slice
    if mode == "allgather":
        ranges, cnt = equal_ranges(n, world.nranks)
    else:
        ranges = row_ranges(n, world.nranks)
    c0, c1 = ranges[world.rank]
    lp, li, ld = csr_row_slice(indptr, indices, data, c0, c1)
    n_local = c1 - c0
    if mode == "allgather":
        n_halo = cnt * world.nranks
        return dict(mode=1, indptr=lp, indices=(li.astype(np.int64) + n_local).astype(np.int32), data=ld,
                    n_local=n_local, n_halo=n_halo, ranges=ranges, send_count=None, recv_count=None,
                    send_idx=None)
    need = needed_columns(li, c0, c1)
    needs = world.allgather_object(need)
    send_count, recv_count, send_idx = halo_plan(world.rank, ranges, needs)
    return dict(mode=0, indptr=lp, indices=localize_columns(li, c0, c1, need), data=ld, n_local=n_local,
                n_halo=int(need.size), ranges=ranges, send_count=send_count, recv_count=recv_count,
                send_idx=send_idx, halo_cols=need)


def partition_host_csr(world, indptr, indices, data, n, mode="halo"):
    """Distribute a (replicated) host CSR matrix: every rank keeps its row block on its GPU."""
    from .linop import CsrOperator
    p = plan_host_csr(world, indptr, indices, data, n, mode)
    op = CsrOperator(p["indptr"], p["indices"], p["data"], (p["n_local"], p["n_local"] + p["n_halo"]))
    op.row_range, op.global_size = tuple(p["ranges"][world.rank]), int(n)
    return attach_exchange(op, p["mode"], p["n_local"], p["n_halo"], p["send_count"], p["recv_count"],
                           p["send_idx"]), p["ranges"]


def partition_row_blocks(world, indptr, indices, data, shape, sliced=False):
    """Row blocks of a (replicated) host m x n CSR matrix for the least-squares solvers (lls/): this rank keeps rows
    ``ranges[rank]`` with their GLOBAL columns.  m-space vectors (rhs, u, r; x of CRAIG-MR) are sliced like the rows,
    n-space vectors (v, w, x) are whole and identical on every rank: ``A * v`` needs no exchange, ``A.T * u``
    (lsqr.py:264) is the local transposed block's product summed over the ranks (`mk_csr_set_row_block`)."""
    from .linop import CsrOperator
    from .sparse import csr_row_slice
    m, n = int(shape[0]), int(shape[1])
    ranges = row_ranges(m, world.nranks)
    r0, r1 = ranges[world.rank]
    lp, li, ld = csr_row_slice(indptr, indices, data, r0, r1)
    op = CsrOperator(lp, li, ld, (r1 - r0, n))
    # mode 1: n-space vectors whole on every rank (A' u all-reduced; right for m >> n).  mode 2 (`sliced`): every rank
    # owns a block of the n-space vectors as well -- A' u is reduce-scattered, the n-space updates and their inner
    # products run on the blocks, only v is all-gathered for the next A v; x is returned whole either way
    _lib.check(_lib.init().mk_csr_set_row_block(op.handle, 2 if sliced else 1))
    op.row_range, op.global_shape = (r0, r1), (m, n)
    return op, ranges


def partition_poisson3d(world, nx, ny, nz, mode="halo", varcoef_seed=None):
    """7-point Poisson matrix on an nx x ny x nz grid, slab-partitioned in z, generated per rank in
    HBM (BASELINE config 5: 512^3 never exists on the host).  `varcoef_seed`: the variable-coefficient
    operator of gallery.poisson3d_varcoef instead (same sparsity, same partition)."""
    from .linop import CsrOperator
    lib = _lib.init()
    n = nx * ny * nz
    if mode == "allgather":
        ranges, cnt = equal_ranges(n, world.nranks)
    else:
        ranges = row_ranges(n, world.nranks, align=nx * ny)
    c0, c1 = ranges[world.rank]
    h = ctypes.c_void_p()
    if varcoef_seed is None:
        _lib.check(lib.mk_csr_poisson3d(nx, ny, nz, c0, c1, ctypes.byref(h)))
    else:
        _lib.check(lib.mk_csr_poisson3d_varcoef(nx, ny, nz, int(varcoef_seed), c0, c1, ctypes.byref(h)))
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    n_local = c1 - c0
    if mode == "allgather":
        n_halo = cnt * world.nranks
        _lib.check(lib.mk_csr_localize(h, 1, c0, c1, n_halo, ctypes.byref(lo), ctypes.byref(hi)))
        op = CsrOperator.from_handle(h.value)
        op.row_range, op.global_size = (c0, c1), n
        return attach_exchange(op, 1, n_local, n_halo), ranges
    _lib.check(lib.mk_csr_localize(h, 0, c0, c1, 0, ctypes.byref(lo), ctypes.byref(hi)))
    op = CsrOperator.from_handle(h.value)
    op.row_range, op.global_size = (c0, c1), n
    windows = world.allgather_object((c0, c1, lo.value, hi.value))
    needs = [banded_needs(*w) for w in windows]
    send_count, recv_count, send_idx = halo_plan(world.rank, ranges, needs)
    return attach_exchange(op, 0, n_local, lo.value + hi.value, send_count, recv_count, send_idx), ranges
