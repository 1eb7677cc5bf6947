"""Build libmikrylov.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pykrylov_amd.build [--force] [--verbose]

The shared object is placed next to this file so that it travels with the source tree.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmikrylov.so")
OBJDIR = os.path.join(HERE, "build")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-fvisibility=hidden",        # only the MK_API entry points of include/mikrylov.h are exported
    "-ffp-contract=off",          # one rounding per multiply and per add, like the NumPy expressions replaced
    *os.environ.get("MK_EXTRA_HIPCC_FLAGS", "").split(),     # (experiments: a second build with other -D switches)
    "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value",
]


def source_files():
    """Every file libmikrylov.so is compiled from, in a fixed order."""
    return (sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.h")))
            + [os.path.join(HERE, "..", "include", "mikrylov.h")])


def source_sha():
    """Digest of the library's sources (names and contents): what mk_build_info() of a current binary returns."""
    import hashlib
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, tag=None, defines=()):
    """`tag` / `defines`: an experiment build beside the product -- libmikrylov_<tag>.so compiled with -D<define> ...,
    loaded instead of the product when MIKRYLOV_LIB points at it (tools/variants.sh)."""
    global OUT, OBJDIR
    if tag:
        OUT = os.path.join(HERE, "libmikrylov_%s.so" % tag)
        OBJDIR = os.path.join(HERE, "build_" + tag)
        FLAGS[:0] = ["-D" + d for d in defines]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "mikrylov.h")]
    os.makedirs(OBJDIR, exist_ok=True)
    objs = []
    procs = []
    sha = source_sha()
    stamp = os.path.join(OBJDIR, "mk_buildinfo.sha")
    sha_changed = not os.path.exists(stamp) or open(stamp).read().strip() != sha
    for src in srcs:
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        info = os.path.basename(src) == "mk_buildinfo.hip"   # the one unit that carries the digest (mk_build_info)
        if force or _newer(obj, [src] + hdrs) or (info and sha_changed):
            cmd = [hipcc] + FLAGS + (['-DMK_SOURCE_SHA="%s"' % sha] if info else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out.strip()):
            sys.stderr.write(out)
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s\n" % src)
    if failed:
        raise RuntimeError("libmikrylov build failed")
    with open(stamp, "w") as fh:
        fh.write(sha + "\n")
    stale = [o for o in glob.glob(os.path.join(OBJDIR, "*.o")) if o not in objs]
    for o in stale:
        os.remove(o)
    if force or procs or _newer(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    _tag = sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else None
    _defs = [a[2:] for a in sys.argv if a.startswith("-D")]
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, tag=_tag, defines=_defs))
