/* oracle/pip_oracle.h -- TEST INFRASTRUCTURE (checker only; never linked into
 * or called from the product path under piplib_amd/).
 *
 * CPU restatement, in plain C, of the PipLib algorithm for the hot path this
 * repository accelerates: traiter()/pivoter() (reference source/traiter.c),
 * the Gomory-cut generator (source/integrer.c), the tableau row store
 * (source/tab.c), the solution tape (source/sol.c) and the pip_solve() front
 * end (source/piplib.c).  Built twice: ORACLE_BITS=64 mirrors the reference's
 * int64 ("dp"/pip64) arithmetic including wrap-around; ORACLE_BITS=128 is the
 * same algorithm on __int128 entries.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * against every .ll golden file of the reference's own test-suite
 * (test/Makefile.am PIPTEST, example/Makefile.am PIPTEST) and against outputs
 * of the reference itself built here (oracle/_ref, see oracle/Makefile).
 */
#ifndef PIP_ORACLE_H
#define PIP_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifndef ORACLE_BITS
#define ORACLE_BITS 64
#endif
#if ORACLE_BITS == 128
typedef __int128 ent;
typedef unsigned __int128 uent;
#else
typedef int64_t ent;
typedef uint64_t uent;
#endif

/* row flags, tab.h:55-62 */
enum { OF_UNIT = 1, OF_PLUS = 2, OF_MINUS = 4, OF_ZERO = 8, OF_CRITIC = 16, OF_UNKNOWN = 32 };
/* solution tape cell kinds, sol.c:42-50 */
enum { OS_FREE = 0, OS_NIL, OS_IF, OS_LIST, OS_FORM, OS_NEW, OS_DIV, OS_VAL, OS_ERROR };
/* traiter flags, funcall.h:32-33 */
enum { OT_INT = 1, OT_DUAL = 2 };

/* status codes of ora_solve_* */
enum {
  ORA_OK = 0,
  ORA_VOID = 1,          /* empty context (front ends print "void") */
  ORA_ERR_OVERFLOW = 2,  /* "Integer overflow", traiter.c:424,442 (exit 1) */
  ORA_ERR_PARMS = 3,     /* "Too much parameters", traiter.c:174,710 */
  ORA_ERR_COLS = 4,      /* "Too many variables", integrer.c:324 (exit 3) */
  ORA_ERR_SOLSIZE = 5,   /* "The solution is too complex", sol.c:97 (exit 26) */
  ORA_ERR_ASSERT = 6,    /* assert(ok_var), integrer.c:499 */
  ORA_ERR_INTERNAL = 7,
  ORA_ERR_NOPIVOT = 8,
  ORA_ERR_BUDGET = 9     /* not the reference's: more constant cuts than the environment's ORACLE_MAX_CUTS allows
                            (the screening of benchmark batches, tests/golden/make_bench_screen.py) */
};

typedef struct ora_cell {
  int kind;
  ent a, b;
} ora_cell;

typedef struct ora ora;

ora *ora_new(void);
void ora_free(ora *o);
void ora_set_deepest_cut(ora *o, int on);
long long ora_pivots(const ora *o);
long long ora_cuts(const ora *o);
int ora_ncells(const ora *o);
const ora_cell *ora_cells(const ora *o);

/* maind.c-style entry: raw PIP tableaux (row-major int64 input).
 * ineq: ni x (nvar+nparm+1), ctx: nc x (nparm+1).  simplify_inputs mirrors the
 * front end's tab_simplify calls when nq != 0 (maind.c:190-196). */
int ora_solve_tableau(ora *o, int nvar, int nparm, int ni, int nc, int bigparm, int nq,
                      const int64_t *ineq, const int64_t *ctx, int simplify_inputs);
/* sol.c:272 (-z option) applied to the tape of the last solve */
void ora_simplify(ora *o);
/* sol_edit text of the last solve (sol.c:291-422) */
void ora_print(const ora *o, FILE *out);

/* whole-file front ends */
int ora_run_dat(FILE *in, FILE *out, int simplify, int deepest); /* pip -s x.dat */
int ora_run_pip(FILE *in, FILE *out);                            /* example < x.pip */
int ora_run_batch(const char *in_path, const char *out_path);    /* batchfmt.h */

#endif
