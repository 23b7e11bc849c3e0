/* examples/solve_small.c -- the C ABI from plain C, no Python anywhere.
 *
 *   gcc examples/solve_small.c -Iinclude -Lpiplib_amd -lpipamd -Wl,-rpath,'$ORIGIN/../piplib_amd' -o examples/solve_small
 *
 * 1. pip_solve() drop-in: the problem of the reference's example/small.pip
 *    (lexmin of (i, j) with i >= 0, j >= 0, i - 3j + 12 >= 0, -2i + j + 3 >= 0, no parameters)
 *    through pipamd_pip_solve; prints the quast exactly as pip_quast_print does.
 * 2. one parametric problem in PIP's native tableau form (reference test/test.dat shape)
 *    through pipamd_solve_tableau; prints the sol_edit text.
 */
#include <stdio.h>
#include <stdlib.h>

#include "piplib_amd.h"

int main(void) {
  pipamd_engine *e = NULL;
  if (pipamd_engine_create(&e, 0)) {
    fprintf(stderr, "no engine: %s\n", pipamd_last_error());
    return 2; /* no GPU: there is no CPU fallback */
  }

  /* ---- 1. PolyLib matrices, as pip_matrix_read would build them ---- */
  long long dom[4][4] = {{1, 1, 0, 0}, {1, 0, 1, 0}, {1, 1, -3, 12}, {1, -2, 1, 3}};
  long long *dom_rows[4] = {dom[0], dom[1], dom[2], dom[3]};
  pipamd_matrix domain = {4, 4, dom_rows, &dom[0][0], 16};
  pipamd_matrix context = {0, 2, NULL, NULL, 0};
  pipamd_options opt = {1, 0, 0, 0, 0, 0, 0, 0}; /* Nq = 1: integer solution */
  pipamd_quast *q = NULL;
  int status = 0;
  int64_t pivots = 0;
  int rc = pipamd_pip_solve(e, &domain, &context, -1, &opt, &q, &status, &pivots);
  if (rc) {
    fprintf(stderr, "pip_solve failed (%d, status %d): %s\n", rc, status, pipamd_last_error());
    return 1;
  }
  char *txt = pipamd_quast_to_string(q, 0);
  printf("%s", txt);
  pipamd_free(txt);
  pipamd_quast_free(q);

  /* ---- 2. PIP tableau form: 2 unknowns, 1 parameter n:  i >= 0.., i + j >= n, i <= 5 ... ---- */
  /* columns: i j | constant | n */
  const int64_t ineq[3 * 4] = {1, 1, 0, -1, /* i + j - n >= 0 */
                               -1, 0, 5, 0, /* 5 - i >= 0     */
                               0, -1, 7, 0 /* 7 - j >= 0     */};
  const int64_t ctx[1 * 2] = {-1, 12}; /* 12 - n >= 0 */
  char *text = NULL;
  rc = pipamd_solve_tableau(e, 2, 1, 3, 1, -1, 1, ineq, ctx, 1, 0, &text, &status, &pivots);
  if (rc) {
    fprintf(stderr, "solve_tableau failed (%d, status %d): %s\n", rc, status, pipamd_last_error());
    return 1;
  }
  printf("%s", text);
  pipamd_free(text);
  pipamd_engine_destroy(e);
  return 0;
}
