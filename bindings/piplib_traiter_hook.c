/* bindings/piplib_traiter_hook.c -- the reference-side binding of the MI355X pivot engine.
 *
 * This is the file a PipLib maintainer adds to the PipLib tree (it includes PipLib's internal
 * header and is compiled with PipLib's own flags).  It defines
 *
 *     void pipamd_traiter_hook_dp(Tableau_dp *tp, Tableau_dp *ctxt,
 *                                 int nvar, int nparm, int ni, int nc, int bigparm, int flags)
 *
 * with the signature and the effect of traiter_dp (source/traiter.c:628, funcall.h:37-47): the
 * answer to the problem (tp, ctxt) is pushed onto the solution tape of source/sol.c.  The pivots
 * run on the GPU behind pipamd_traiter() (include/piplib_amd.h); the cells it returns are replayed
 * through sol_nil / sol_if / sol_list / sol_forme / sol_new / sol_div / sol_val (sol.c:104-209), so
 * everything downstream of traiter -- sol_simplify, sol_edit, sol_quast_edit, pip_quast_print --
 * is the reference's own code, unchanged.
 *
 * PipLib is switched over by compiling its callers of traiter with
 *     -Dtraiter_dp=pipamd_traiter_hook_dp
 * (source/piplib.c: the empty-context test at :823 and the main call at :858; a maind.c-style
 * driver likewise) and linking this file and -lpipamd.  oracle/Makefile does exactly that with the
 * reference sources where they lie (oracle/_ref/libpiplib_gpu_dp.so, oracle/_ref/refpip_gpu) and
 * the -m gpu tests run the reference's test (.dat) and example (.pip) suites through it.
 *
 * Flavour-generic, like PipLib itself (PIPLIB_NAME, include/piplib/piplib.h:40-88, source/funcall.h:37-47): compiled
 * with -DPIPLIB_INT_DP it defines pipamd_traiter_hook_dp over the 64-bit engine (pipamd_traiter); compiled with
 * -DPIPLIB_INT_GMP it defines pipamd_traiter_hook_gmp over the 128-bit engine (pipamd_traiter128): the overflow-safe
 * device flavour serves the arbitrary-precision build as long as the input coefficients fit 64 bits (they come from a
 * PipMatrix / a .dat file) -- answers whose numbers fit 128 bits come back exact, a problem that would leave 128 bits
 * ends with the engine's "Integer overflow" like the fixed-width flavours (oracle/_ref/libpiplib_gpu_gmp.so,
 * oracle/_ref/refpip_gpu_gmp; tests/test_gpu_golden.py runs the reference's .dat suite through it).
 */
#include <stdio.h>
#include <stdlib.h>

#include "pip.h" /* PipLib's internal header (source/pip.h): Tableau_xx, Index/Denom/Flag, sol_*_xx */
#include "piplib_amd.h"

long long pipamd_hook_pivots; /* pivots of the calls so far (drivers print it) */
extern int deepest_cut_xx;     /* source/piplib.c:53 (set from PipOptions.Deepest_cut, piplib.c:746) */

#define pipamd_traiter_hook_xx PIPLIB_NAME(pipamd_traiter_hook)
#ifdef PIPLIB_INT_GMP
typedef pipamd_sol_cell128 hook_cell;
/* an Entier of the caller as int64 (input coefficients) */
static long long entier_in(piplib_int_t_xx v) {
  if (!mpz_fits_slong_p(v)) {
    fprintf(stderr, "piplib (GPU traiter): an input coefficient does not fit 64 bits\n");
    exit(1);
  }
  return mpz_get_si(v);
}
/* a 128-bit (low, high) value of the engine as an mpz */
static void entier_out(mpz_t r, int64_t lo, int64_t hi) {
  unsigned long long w[2];
  const int neg = hi < 0;
  unsigned __int128 m = ((unsigned __int128)(unsigned long long)hi << 64) | (unsigned long long)lo;
  if (neg) m = (unsigned __int128)0 - m;
  w[0] = (unsigned long long)m;
  w[1] = (unsigned long long)(m >> 64);
  mpz_import(r, 2, -1, sizeof w[0], 0, 0, w);
  if (neg) mpz_neg(r, r);
}
#else
typedef pipamd_sol_cell hook_cell;
#define entier_in(v) ((long long)(v))
#endif

static pipamd_engine *hook_engine(void) {
  static pipamd_engine *eng;
  if (!eng) {
    const char *dev = getenv("PIPAMD_DEVICE");
    if (pipamd_engine_create(&eng, dev ? atoi(dev) : 0) != PIPAMD_OK) {
      fprintf(stderr, "piplib (GPU traiter): %s\n", pipamd_last_error());
      exit(1); /* no GPU: an error, never a CPU fallback */
    }
  }
  return eng;
}

void pipamd_traiter_hook_xx(Tableau_xx *tp, Tableau_xx *ctxt, int nvar, int nparm, int ni, int nc, int bigparm,
                            int flags) {
  const int ncol = nvar + nparm + 1;
  long long *rows = malloc(sizeof(long long) * ((size_t)ni * ncol + 1));
  long long *crow = malloc(sizeof(long long) * ((size_t)nc * (nparm + 1) + 1));
  hook_cell *cells = NULL;
  size_t n = 0, k;
  int i, j, status = 0, rc;
  int64_t piv = 0;
  if (!rows || !crow) {
    fprintf(stderr, "Memory overflow\n");
    exit(1);
  }
  /* the callers build tp with tab_Matrix2Tableau / tab_get / expanser: nvar unit rows, then ni
   * Unknown rows with denominator 1 (tab.c:222-248, 292-393) */
  for (i = 0; i < ni; i++) {
    if (Flag(tp, nvar + i) != Unknown || !piplib_int_one(Denom(tp, nvar + i))) {
      fprintf(stderr, "piplib (GPU traiter): row %d is not a fresh Unknown row\n", i);
      exit(1);
    }
    for (j = 0; j < ncol; j++) rows[(size_t)i * ncol + j] = entier_in(Index(tp, nvar + i, j));
  }
  for (i = 0; i < nc; i++)
    for (j = 0; j <= nparm; j++) crow[(size_t)i * (nparm + 1) + j] = entier_in(Index(ctxt, i, j));
#ifdef PIPLIB_INT_GMP
  rc = pipamd_traiter128(hook_engine(), nvar, nparm, ni, nc, bigparm, flags & (TRAITER_INT | TRAITER_DUAL), deepest_cut_xx,
                         (const int64_t *)rows, (const int64_t *)crow, &cells, &n, &status, &piv);
#else
  rc = pipamd_traiter(hook_engine(), nvar, nparm, ni, nc, bigparm, flags & (TRAITER_INT | TRAITER_DUAL), deepest_cut_xx,
                      (const int64_t *)rows, (const int64_t *)crow, &cells, &n, &status, &piv);
#endif
  pipamd_hook_pivots += piv;
  free(rows);
  free(crow);
  if (rc == PIPAMD_E_SOLVER && status == PIPAMD_ST_OVERFLOW) {
    fprintf(stderr, "Integer overflow\n"); /* traiter.c:424,442 */
    exit(1);
  }
  if (rc != PIPAMD_OK) {
    fprintf(stderr, "piplib (GPU traiter): %s (status %d)\n", pipamd_last_error(), status);
    exit(1);
  }
#ifdef PIPLIB_INT_GMP
  {
    mpz_t a, b;
    mpz_init(a);
    mpz_init(b);
    for (k = 0; k < n; k++) switch (cells[k].kind) {
        case PIPAMD_SOL_NIL: sol_nil_xx(); break;
        case PIPAMD_SOL_IF: sol_if_xx(); break;
        case PIPAMD_SOL_LIST: sol_list_xx((int)cells[k].param1_lo); break;
        case PIPAMD_SOL_FORM: sol_forme_xx((int)cells[k].param1_lo); break;
        case PIPAMD_SOL_NEW: sol_new_xx((int)cells[k].param1_lo); break;
        case PIPAMD_SOL_DIV: sol_div_xx(); break;
        case PIPAMD_SOL_VAL:
          entier_out(a, cells[k].param1_lo, cells[k].param1_hi);
          entier_out(b, cells[k].param2_lo, cells[k].param2_hi);
          sol_val_xx(a, b);
          break;
        default: fprintf(stderr, "piplib (GPU traiter): unknown tape cell %d\n", cells[k].kind); exit(1);
      }
    mpz_clear(a);
    mpz_clear(b);
  }
#else
  for (k = 0; k < n; k++) switch (cells[k].kind) {
      case PIPAMD_SOL_NIL: sol_nil_xx(); break;
      case PIPAMD_SOL_IF: sol_if_xx(); break;
      case PIPAMD_SOL_LIST: sol_list_xx((int)cells[k].param1); break;
      case PIPAMD_SOL_FORM: sol_forme_xx((int)cells[k].param1); break;
      case PIPAMD_SOL_NEW: sol_new_xx((int)cells[k].param1); break;
      case PIPAMD_SOL_DIV: sol_div_xx(); break;
      case PIPAMD_SOL_VAL: sol_val_xx(cells[k].param1, cells[k].param2); break;
      default: fprintf(stderr, "piplib (GPU traiter): unknown tape cell %d\n", cells[k].kind); exit(1);
    }
#endif
  pipamd_free(cells);
}
