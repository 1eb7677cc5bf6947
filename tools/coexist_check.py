"""Does libmikrylov coexist with PyTorch's bundled HIP runtime in one process?  (GPU box only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch-first"
import numpy as np


def mk():
    from pykrylov_amd import CG, gallery, _lib
    _lib.init(0)
    op = gallery.poisson2d(64)
    x = np.ones(64 * 64)
    rhs = op * x
    s = CG(op)
    s.solve(rhs)
    return s.nMatvec, float(np.abs(s.x - 1).max())


def th():
    import torch
    a = torch.ones(1000, device="cuda", dtype=torch.float64)
    return float((a * 2).sum().item()), torch.version.hip


if order == "torch-first":
    print("torch:", th()); print("mk:", mk()); print("torch again:", th())
else:
    print("mk:", mk()); print("torch:", th()); print("mk again:", mk())
loaded = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "librccl" in l or "libmikrylov" in l))
print("\n".join(loaded))
