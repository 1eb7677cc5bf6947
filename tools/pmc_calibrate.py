"""Run known-traffic kernels (4/8/16 B per lane, read and write, 1 GiB each, far beyond the 256 MiB
Infinity Cache) so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be scaled to bytes on gfx950."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pykrylov_amd import _lib

lib = _lib.init(0)
nbytes = 1 << 30
buf = _lib.DeviceArray(nbytes // 8)
for write in (0, 1):
    for width in (4, 8, 16):
        for _ in range(3):
            _lib.check(lib.mk_calib_stream(buf.ptr, nbytes, width, write))
print("calibration kernels done: %d bytes per launch" % nbytes)
