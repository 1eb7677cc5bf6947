// temporary: solvers not yet implemented
#include "mk_solver.h"
mk_solver *mk_make_symmlq() { return nullptr; }
