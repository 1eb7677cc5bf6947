"""One rank's share of the 8-GPU run, executed on ONE GPU: rank 3's slab of the 512^3 problem (rows [3n/8, 4n/8), 16.8 M
rows, columns localised to [own | lower plane | upper plane]) with the production multi-GPU code path -- pack kernel,
interior launch, exchange, boundary launch, all-reduce call sites, both streams -- through the host-staged transport
in a LOOPBACK: this process plays all eight ranks' bookkeeping, the planes it would send to ranks 2 and 4 come back as
the planes it would receive from them (a z-periodic slab: still symmetric positive definite), the all-reduce adds
nothing.  The per-kernel times (rocprofv3 --kernel-trace, tools/slab_budget.sh) are what an 8-GPU pass costs per rank in
kernels; the messages themselves (2 x 2 MiB per neighbour over xGMI) and the RCCL call layer are NOT measured here.

    python tools/slab_budget.py [varcoef|const] [passes]
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pykrylov_amd import _lib, dist  # noqa: E402
from pykrylov_amd.generic import DeviceRun  # noqa: E402

NR, RANK, M = 8, 3, 512


class LoopbackWorld(dist.World):
    """Rank RANK of NR with nobody else there: planning metadata is computed instead of gathered."""

    def __init__(self, ranges, plane):
        dist.World.__init__(self, RANK, NR, None)
        self.ranges, self.plane = ranges, plane

    def allgather_object(self, obj):
        assert isinstance(obj, tuple) and len(obj) == 4, obj        # (c0, c1, halo_lo, halo_hi) of partition_poisson3d
        return [(c0, c1, self.plane if r > 0 else 0, self.plane if r < NR - 1 else 0)
                for r, (c0, c1) in enumerate(self.ranges)]


def loopback_callbacks():
    def view(ptr, count):
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(count,))

    def allreduce(buf, count):
        return 0                                             # (the other ranks contribute nothing)

    def exchange(send, send_count, send_off, recv, recv_count, recv_off):
        # what goes to rank 4 (my top plane) arrives as if from rank 2 (below me), and vice versa
        for src, dst in ((RANK + 1, RANK - 1), (RANK - 1, RANK + 1)):
            cnt = send_count[src]
            assert cnt == recv_count[dst]
            view(recv, recv_off[dst] + cnt)[recv_off[dst]:] = view(send, send_off[src] + cnt)[send_off[src]:]
        return 0

    def allgather(send, count, recv):
        return 1
    return (_lib.HOST_ALLREDUCE_FN(allreduce), _lib.HOST_EXCHANGE_FN(exchange), _lib.HOST_ALLGATHER_FN(allgather))


def build_slab(kind):
    lib = _lib.init(0)
    n = M ** 3
    ranges = dist.row_ranges(n, NR, align=M * M)
    world = LoopbackWorld(ranges, M * M)
    world._cbs = loopback_callbacks()
    _lib.check(lib.mk_comm_init_host(NR, RANK, *world._cbs))
    op, _ = dist.partition_poisson3d(world, M, M, M, mode="halo", varcoef_seed=7 if kind == "varcoef" else None)
    return lib, world, op


def slab_info(lib, op):
    fmt, chunks, nd = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    tiles, mb = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.mk_csr_format_info(op.handle, ctypes.byref(fmt), ctypes.byref(tiles), ctypes.byref(chunks),
                                      ctypes.byref(nd), ctypes.byref(mb)))
    ni, nb = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(lib.mk_csr_overlap_info(op.handle, ctypes.byref(ni), ctypes.byref(nb)))
    return dict(format=fmt.value, tiles_windowed=tiles.value, matrix_bytes_per_product=mb.value,
                tiles_interior=ni.value, tiles_boundary=nb.value, rows=int(op.local_size), halo=int(op.halo_size),
                nnz=int(op.nnz))


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "varcoef"
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    lib, world, op = build_slab(kind)
    info = slab_info(lib, op)
    n_local = op.local_size
    # per-rank HBM footprint: CSR arrays kept resident + format data + the solver's vectors
    csr_bytes = 12 * info["nnz"] + 4 * (n_local + 1)
    vec_bytes = 8 * (2 * (n_local + op.halo_size) + 2 * n_local)
    info["hbm_footprint_bytes"] = {"csr_arrays": csr_bytes, "format_data": info["matrix_bytes_per_product"],
                                   "cg_vectors": vec_bytes, "total": csr_bytes + info["matrix_bytes_per_product"] + vec_bytes}
    ones = _lib.DeviceArray.from_numpy(np.ones(op.shape[1]))
    rhs = _lib.DeviceArray(n_local)
    op.spmv_device(ones.ptr, rhs.ptr)
    run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=1 << 60, check_curvature=0)
    run.setup()
    assert run.iterate(20) == 20
    _lib.check(lib.mk_sync())
    t0 = time.perf_counter()
    assert run.iterate(passes) == passes
    _lib.check(lib.mk_sync())
    info["wall_ms_per_pass_with_host_staged_loopback"] = 1e3 * (time.perf_counter() - t0) / passes
    res = run.finish()
    info["residual_finite"] = bool(np.isfinite(res.residNorm))
    info["kind"], info["passes"] = kind, passes
    print(json.dumps(info))
    run.close()
    op.free()
    lib.mk_comm_destroy()


if __name__ == "__main__":
    main()
