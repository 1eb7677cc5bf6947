// temporary: solvers not yet implemented
#include "mk_solver.h"
mk_solver *mk_make_bicgstab() { return nullptr; }
mk_solver *mk_make_cgs() { return nullptr; }
mk_solver *mk_make_tfqmr() { return nullptr; }
mk_solver *mk_make_minres() { return nullptr; }
mk_solver *mk_make_symmlq() { return nullptr; }
