// piplib_amd/csrc/pip_tree.cpp -- layer 3 (placeholder until the host decision tree lands).
#include "pip_host.h"

extern "C" int pipamd_solve_tableau(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int nq,
                                    const int64_t *ineq, const int64_t *ctx, int simplify, int deepest_cut, char **text,
                                    int *status, int64_t *pivots) {
  (void)e; (void)nvar; (void)nparm; (void)ni; (void)nc; (void)bigparm; (void)nq; (void)ineq; (void)ctx;
  (void)simplify; (void)deepest_cut; (void)text; (void)status; (void)pivots;
  pipamd_set_error("pipamd_solve_tableau: not built yet");
  return PIPAMD_E_INVALID;
}
