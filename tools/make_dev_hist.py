"""Device-generated fixture for bench.py's N > 1 self-validation (SURVEY.md 8e "P > 1 vs P = 1"): the residual histories
of the first 60 CG passes (rhs = A 1, x0 = 0, tolerances 0) of the 3-D Poisson workloads as ONE MI355X produces them.
Run on the GPU box:   python tools/make_dev_hist.py gpurun_out/dev_hist_512.npz   and copy the file to tests/golden/.
The file holds numbers only: inputs are fully described by (grid, seed), outputs are the histories."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pykrylov_amd import _lib, gallery  # noqa: E402
from pykrylov_amd.generic import DeviceRun  # noqa: E402

_lib.init(0)
out = {"passes": np.int64(bench.PARITY_PASSES), "seed": np.int64(bench.VARCOEF_SEED),
       "generated_on": np.array(_lib.device_info()["name"])}
for m in (64, 512):
    for tag, op in (("poisson3d-%d" % m, gallery.poisson3d(m)),
                    ("poisson3d-%d-varcoef" % m, gallery.poisson3d_varcoef(m, seed=bench.VARCOEF_SEED))):
        n = m ** 3
        ones = _lib.DeviceArray.from_numpy(np.ones(n))
        rhs = _lib.DeviceArray(n)
        op.spmv_device(ones.ptr, rhs.ptr)
        hists = []
        for rep in range(2):                                   # twice: the run is deterministic, bit for bit
            run = DeviceRun(op, _lib.MK_CG, rhs, None, abstol=0.0, reltol=0.0, matvec_max=bench.PARITY_PASSES,
                            check_curvature=1)
            res = run.run()
            assert res.nMatvec == bench.PARITY_PASSES
            hists.append(run.history())
            run.close()
        assert np.array_equal(hists[0], hists[1])
        out["hist_" + tag.replace("-", "_")] = hists[0]
        print(tag, hists[0][0], hists[0][-1], flush=True)
        for b in (ones, rhs):
            b.free()
        op.free()
np.savez(sys.argv[1], **out)
