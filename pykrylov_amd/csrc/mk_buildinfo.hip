// mk_buildinfo.hip -- ties the shared object to the source tree it was compiled from.
//
// pykrylov_amd/build.py hashes every file the library is compiled from (csrc/*.hip, csrc/*.h, include/mikrylov.h) and compiles
// the digest into this translation unit (-DMK_SOURCE_SHA=...; the only one that sees it, so an edit elsewhere recompiles
// that file and this one).  pykrylov_amd/_lib.py compares mk_build_info() with the digest of the tree it finds itself in and
// refuses a stale binary: libmikrylov.so is git-ignored and travels prebuilt to the GPU box.
#include "mk_internal.h"

#ifndef MK_SOURCE_SHA
#define MK_SOURCE_SHA "unknown"
#endif

extern "C" const char *mk_build_info(void) { return MK_SOURCE_SHA; }
