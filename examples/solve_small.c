/* examples/solve_small.c -- the C ABI from plain C, no Python anywhere.
 *
 *   gcc examples/solve_small.c -Iinclude -Lpiplib_amd -lpipamd -Wl,-rpath,'$ORIGIN/../piplib_amd' -o examples/solve_small
 *
 * 1. pipamd_traiter: the tableau the reference's tab_Matrix2Tableau builds for example/small.pip
 *    (lexmin of (i, j) >= 0 with i - 3j + 12 >= 0 and -2i + j + 3 >= 0, no parameters), integer solve.
 * 2. pipamd_solve_tableau: one parametric problem in PIP's native tableau form with a context row
 *    (maind.c semantics: tab_simplify, empty-context test, traiter).
 * Both print the solution tape, one cell per line: "cell <kind> <param1> <param2>" -- what sol.c
 * would hold after traiter(); bindings/piplib_traiter_hook.c replays such cells into the reference.
 */
#include <stdio.h>
#include <stdlib.h>

#include "piplib_amd.h"

static void print_tape(const char *what, const pipamd_sol_cell *c, size_t n, long long pivots) {
  printf("tape %s: %zu cells, %lld pivots\n", what, n, pivots);
  for (size_t i = 0; i < n; i++)
    printf("cell %d %lld %lld\n", c[i].kind, (long long)c[i].param1, (long long)c[i].param2);
}

int main(void) {
  pipamd_engine *e = NULL;
  pipamd_sol_cell *cells = NULL;
  size_t n = 0;
  int status = 0, rc;
  int64_t pivots = 0;
  if (pipamd_engine_create(&e, 0)) {
    fprintf(stderr, "no engine: %s\n", pipamd_last_error());
    return 2; /* no GPU: there is no CPU fallback */
  }

  /* ---- 1. traiter(): columns i j | constant ---- */
  const int64_t small[2 * 3] = {1, -3, 12, /* i - 3j + 12 >= 0 */
                                -2, 1, 3 /* -2i + j + 3 >= 0 */};
  rc = pipamd_traiter(e, 2, 0, 2, 0, -1, PIPAMD_T_INT, 0, small, NULL, &cells, &n, &status, &pivots);
  if (rc) {
    fprintf(stderr, "traiter failed (%d, status %d): %s\n", rc, status, pipamd_last_error());
    return 1;
  }
  print_tape("small", cells, n, (long long)pivots);
  pipamd_free(cells);

  /* ---- 2. PIP tableau form: 2 unknowns, 1 parameter n; columns i j | constant | n ---- */
  const int64_t ineq[3 * 4] = {1, 1, 0, -1, /* i + j - n >= 0 */
                               -1, 0, 5, 0, /* 5 - i >= 0     */
                               0, -1, 7, 0 /* 7 - j >= 0     */};
  const int64_t ctx[1 * 2] = {-1, 12}; /* 12 - n >= 0 */
  rc = pipamd_solve_tableau(e, 2, 1, 3, 1, -1, 1, ineq, ctx, 1, 0, &cells, &n, &status, &pivots);
  if (rc) {
    fprintf(stderr, "solve_tableau failed (%d, status %d): %s\n", rc, status, pipamd_last_error());
    return 1;
  }
  print_tape("parametric", cells, n, (long long)pivots);
  pipamd_free(cells);
  pipamd_engine_destroy(e);
  return 0;
}
