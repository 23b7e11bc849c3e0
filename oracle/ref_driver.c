/* oracle/ref_driver.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A small driver of our own around the *real* reference library
 * (oracle/_ref/libpiplib_ref_dp.so, compiled by oracle/Makefile straight from
 * /root/reference/source/{piplib,traiter,integrer,tab,sol}.c with
 * -DPIPLIB_INT_DP, i.e. the reference's "pip64"/piplib64 build).
 *
 * Flavour-generic: every reference name is spelled through the reference's own *_xx aliases
 * (PIPLIB_NAME, include/piplib/piplib.h:40-88, source/funcall.h:37-47), so the same file compiled
 * with -DPIPLIB_INT_GMP drives the arbitrary-precision build (oracle/_ref/refpip_gmp over
 * libpiplib_ref_gmp.so: the authority for the 128-bit Entier engine beyond 2^63).  That build also
 * interposes libgmp's mpz_mul / mpz_add / mpz_sub and reports, per problem, the widest value the
 * reference produced in a tableau / cut / context computation and the widest determinant
 * (batch_res.reserved), so that a fixture knows where a 128-bit run could wrap.
 *
 * The reference's own command-line front end (source/maind.c) cannot be built
 * here because it includes a generated "version.h"; this file replaces it with
 * the minimum needed to (1) turn a .dat file into the .ll text the reference
 * test-suite diffs against (test/Makefile.am:62-87), (2) run the .pip examples
 * through pip_solve (example/example.c), and (3) solve a binary batch of
 * tableaux while counting calls to pivoter_xx, for the CPU baseline.
 *
 * Modes
 *   refpip dat  <in.dat>            -> .ll text on stdout           [-z simplify]
 *   refpip pip  < in.pip            -> example.c-style text on stdout
 *   refpip batch <in.bin> <out.bin> -> binary batch (format: oracle/batchfmt.h)
 *
 * The reference aborts with exit() on overflow ("Integer overflow",
 * traiter.c:424-427,441-444).  In batch mode we interpose exit() (this is the
 * main executable, linked -rdynamic, so its definition wins) and longjmp back
 * so that one overflowing tableau does not lose the rest of the batch.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

#include "pip.h" /* the reference's internal header, found via -I/root/reference/source */
#include "batchfmt.h"

extern int verbose_xx;
extern int deepest_cut_xx;
#ifndef pivoter_xx
#define pivoter_xx PIPLIB_NAME(pivoter) /* the alias is private to traiter.c:344 */
#endif
#define STR_(x) #x
#define STR(x) STR_(x)

/* ---- pivot counter: interposes the reference's pivoter_xx (traiter.c:345) ----
 * Not in the REF_NO_COUNT build (oracle/_ref/refpip_fast: this driver and the five reference
 * sources in one -O3 executable, the reference's calls bound directly): that one is only timed,
 * it reports 0 pivots and the caller takes the counts from a pass of the counting build. */
static long long g_pivots;
#ifdef REF_GPU_HOOK
/* oracle/_ref/refpip_gpu: traiter_dp is bound to bindings/piplib_traiter_hook.c (-Dtraiter_dp=...),
 * the pivots run on the GPU and the hook counts them */
extern long long pipamd_hook_pivots;
#define g_pivots pipamd_hook_pivots
#elif !defined(REF_NO_COUNT)
static int (*real_pivoter)(Tableau_xx *, int, int, int, int);
#ifdef PIPLIB_INT_GMP
static const void *g_det;                 /* the determinant of the tableau being pivoted (tab.h:76-81) */
static unsigned g_entry_bits, g_det_bits; /* widest results since the last reset */
#endif
int pivoter_xx(Tableau_xx *tp, int pivi, int nvar, int nparm, int ni) {
  if (!real_pivoter)
    real_pivoter = (int (*)(Tableau_xx *, int, int, int, int))dlsym(RTLD_NEXT, STR(pivoter_xx));
  g_pivots++;
#ifdef PIPLIB_INT_GMP
  g_det = (const void *)tp->determinant;
#endif
  return real_pivoter(tp, pivi, nvar, nparm, ni);
}
#endif

#if defined(PIPLIB_INT_GMP) && !defined(REF_GPU_HOOK) && !defined(REF_NO_COUNT)
/* ---- width tracker: the reference's arithmetic goes through libgmp's PLT entries; the main
 * executable's definitions win, record the width of the result and call the real function ---- */
static void track(mpz_srcptr r) {
  const unsigned b = mpz_sgn(r) ? (unsigned)mpz_sizeinbase(r, 2) : 0;
  if ((const void *)r == g_det) {
    if (b > g_det_bits) g_det_bits = b;
  } else if (b > g_entry_bits)
    g_entry_bits = b;
}
#define TRACKED(name)                                                        \
  void __gmpz_##name(mpz_ptr r, mpz_srcptr a, mpz_srcptr b) {                \
    static void (*real)(mpz_ptr, mpz_srcptr, mpz_srcptr);                    \
    if (!real) real = (void (*)(mpz_ptr, mpz_srcptr, mpz_srcptr))dlsym(RTLD_NEXT, "__gmpz_" #name); \
    real(r, a, b);                                                           \
    track(r);                                                                \
  }
TRACKED(mul)
TRACKED(add)
TRACKED(sub)
#endif

/* ---- exit() trap ---- */
static jmp_buf g_trap;
static int g_trap_armed;
static int g_trap_code;
void exit(int code) {
  if (g_trap_armed) {
    g_trap_armed = 0;
    g_trap_code = code;
    longjmp(g_trap, 1);
  }
  fflush(NULL);
  _exit(code);
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* Solve one already-built (ineq, context) pair the way the reference front
 * ends do (maind.c:196-231 / piplib.c:813-871): empty-context test first, then
 * the main traiter call.  Returns 1 if a solution tree was produced at *xq. */
static int run_traiter(Tableau_xx *ineq, Tableau_xx *context, int nvar, int nparm, int ni, int nc,
                       int bigparm, int nq, int p) {
  int non_vide = 1;
  if (nc) {
    Tableau_xx *ctxt = expanser_xx(context, nparm, nc, nparm + 1, nparm, 0, 0);
    traiter_xx(ctxt, NULL, nparm, 0, nc, 0, -1, TRAITER_INT);
    non_vide = is_not_Nil_xx(p);
    sol_reset_xx(p);
  }
  if (non_vide) traiter_xx(ineq, context, nvar, nparm, ni, nc, bigparm, nq ? TRAITER_INT : 0);
  return non_vide;
}

/* ------------------------------------------------------------------ dat mode */
static int read_int(FILE *in, int *v) {
  piplib_int_t_xx x;
  int rc = 0;
  piplib_int_init(x);
  if (dscanf_xx(in, &x) < 0)
    rc = -1;
  else
    *v = piplib_int_get_si(x);
  piplib_int_clear(x);
  return rc;
}

static int mode_dat(const char *path, int simplify) {
  FILE *in = fopen(path, "r");
  FILE *out = stdout;
  int c;
  if (!in) {
    fprintf(stderr, "%s unaccessible\n", path);
    return 1;
  }
  verbose_xx = -1;
  sol_init_xx();
  tab_init_xx();
  while ((c = dgetc_xx(in)) != EOF) {
    int nvar, nparm, ni, nc, bigparm, nq, level = 0, p, xq, q;
    struct high_water_mark_xx hq;
    Tableau_xx *ineq, *context;
    if (c != '(') continue;
    /* echo the comment group, as the reference front end does */
    fputc('(', out);
    while ((c = dgetc_xx(in)) != EOF) {
      if (c == '(') level++;
      else if (c == ')' && --level == 0) break;
      fputc(c, out);
    }
    if (read_int(in, &nvar) || read_int(in, &nparm) || read_int(in, &ni) || read_int(in, &nc) ||
        read_int(in, &bigparm) || read_int(in, &nq)) {
      fprintf(out, "\nSyntax error\n)\n");
      break;
    }
    hq = tab_hwm_xx();
    ineq = tab_get_xx(in, ni, nvar + nparm + 1, nvar);
    if (!ineq) break;
    if (nq) tab_simplify_xx(ineq, nvar);
    context = tab_get_xx(in, nc, nparm + 1, 0);
    if (!context) break;
    if (nq) tab_simplify_xx(context, nparm);
    xq = p = sol_hwm_xx();
    if (run_traiter(ineq, context, nvar, nparm, ni, nc, bigparm, nq, p)) {
      fputs(")\n", out);
      if (simplify) sol_simplify_xx(xq);
      q = sol_hwm_xx();
      while ((xq = sol_edit_xx(out, xq)) != q)
        ;
      sol_reset_xx(p);
    } else
      fprintf(out, "void\n");
    tab_reset_xx(hq);
    fprintf(out, ")\n");
    fflush(out);
  }
  fclose(in);
  return 0;
}

/* ------------------------------------------------------------------ pip mode */
static int mode_pip(void) {
  /* Same stdin protocol and stdout text as example/example.c:72-118. */
  int bignum;
  char s[1024];
  PipMatrix_xx *domain, *context;
  PipQuast_xx *solution;
  PipOptions_xx *options;
  printf("[PIP2-like future input] Please enter:\n- the context matrix,\n");
  context = pip_matrix_read_xx(stdin);
  pip_matrix_print_xx(stdout, context);
  printf("- the bignum column (start at 0, -1 if no bignum),\n");
  if (fscanf(stdin, " %d", &bignum) != 1) return 1;
  printf("%d\n", bignum);
  printf("- the constraint matrix.\n");
  domain = pip_matrix_read_xx(stdin);
  pip_matrix_print_xx(stdout, domain);
  printf("\n");
  options = pip_options_init_xx();
  while (fgets(s, sizeof s, stdin)) {
    if (!strncasecmp(s, "Maximize", 8)) options->Maximize = 1;
    if (!strncasecmp(s, "Urs_parms", 9)) options->Urs_parms = 1;
    if (!strncasecmp(s, "Urs_unknowns", 12)) options->Urs_unknowns = 1;
    if (!strncasecmp(s, "Rational", 8)) options->Nq = 0;
    if (!strncasecmp(s, "Dual", 4)) options->Compute_dual = 1;
  }
  if (bignum > 0) bignum += domain->NbColumns - context->NbColumns;
  solution = pip_solve_xx(domain, context, bignum, options);
  pip_quast_print_xx(stdout, solution, 0);
  fprintf(stderr, "pivots %lld\n", g_pivots);
  return 0;
}

/* ---------------------------------------------------------------- batch mode */
static int mode_batch(const char *in_path, const char *out_path) {
  FILE *in = fopen(in_path, "rb"), *out = fopen(out_path, "wb");
  struct batch_hdr bh;
  struct batch_out_hdr oh;
  double t_solve = 0;
  long long total_pivots = 0;
  unsigned k;
  if (!in || !out) return 2;
  if (fread(&bh, sizeof bh, 1, in) != 1 || bh.magic != BATCH_MAGIC) return 3;
  memset(&oh, 0, sizeof oh);
  oh.magic = BATCH_MAGIC;
  oh.count = bh.count;
  fwrite(&oh, sizeof oh, 1, out);
  verbose_xx = -1;
  deepest_cut_xx = (bh.flags & BATCH_F_DEEPEST) ? 1 : 0;
  sol_init_xx();
  tab_init_xx();
  for (k = 0; k < bh.count; k++) {
    struct batch_prob ph;
    struct batch_res rh;
    struct high_water_mark_xx hq;
    Tableau_xx *ineq, *context;
    long long *buf;
    char *txt = NULL;
    size_t txtlen = 0;
    int i, j, ncol, p, xq, q;
    double t0;
    if (fread(&ph, sizeof ph, 1, in) != 1) return 4;
    ncol = ph.nvar + ph.nparm + 1;
    buf = malloc(sizeof(long long) * ((size_t)ph.ni * ncol + (size_t)ph.nc * (ph.nparm + 1) + 1));
    if (fread(buf, sizeof(long long), (size_t)ph.ni * ncol + (size_t)ph.nc * (ph.nparm + 1), in) !=
        (size_t)ph.ni * ncol + (size_t)ph.nc * (ph.nparm + 1))
      return 5;
    memset(&rh, 0, sizeof rh);
    hq = tab_hwm_xx();
    xq = p = sol_hwm_xx();
    g_pivots = 0;
#if defined(PIPLIB_INT_GMP) && !defined(REF_GPU_HOOK) && !defined(REF_NO_COUNT)
    g_entry_bits = g_det_bits = 0;
    g_det = NULL;
#endif
    /* tab_get_xx's effect (tab.c:222-248) without the text parsing: rows are
     * Unknown with denominator 1. */
    ineq = tab_alloc_xx(ph.ni, ncol, ph.nvar);
    for (i = 0; i < ph.ni; i++) {
      Flag(ineq, ph.nvar + i) = Unknown;
      piplib_int_set_si(Denom(ineq, ph.nvar + i), 1);
      for (j = 0; j < ncol; j++) piplib_int_set_si(Index(ineq, ph.nvar + i, j), buf[(size_t)i * ncol + j]);
    }
    context = tab_alloc_xx(ph.nc, ph.nparm + 1, 0);
    for (i = 0; i < ph.nc; i++) {
      Flag(context, i) = Unknown;
      piplib_int_set_si(Denom(context, i), 1);
      for (j = 0; j <= ph.nparm; j++)
        piplib_int_set_si(Index(context, i, j), buf[(size_t)ph.ni * ncol + (size_t)i * (ph.nparm + 1) + j]);
    }
    t0 = now_s();
    g_trap_armed = 1;
    if (setjmp(g_trap) == 0) {
      int nv;
      if (ph.nq && !(bh.flags & BATCH_F_NOSIMPLIFY)) {
        tab_simplify_xx(ineq, ph.nvar);
        tab_simplify_xx(context, ph.nparm);
      }
      nv = run_traiter(ineq, context, ph.nvar, ph.nparm, ph.ni, ph.nc, ph.bigparm, ph.nq, p);
      g_trap_armed = 0;
      t_solve += now_s() - t0;
      rh.status = nv ? BATCH_ST_OK : BATCH_ST_VOID;
      if (nv && !(bh.flags & BATCH_F_NOTEXT)) {
        FILE *ms = open_memstream(&txt, &txtlen);
        q = sol_hwm_xx();
        while ((xq = sol_edit_xx(ms, xq)) != q)
          ;
        fclose(ms);
      }
    } else {
      t_solve += now_s() - t0;
      rh.status = BATCH_ST_ABORT;
      rh.abort_code = g_trap_code;
    }
    rh.pivots = g_pivots;
    total_pivots += g_pivots;
    rh.text_len = (unsigned)txtlen;
#if defined(PIPLIB_INT_GMP) && !defined(REF_GPU_HOOK) && !defined(REF_NO_COUNT)
    rh.reserved = (g_entry_bits > 0xffff ? 0xffffu : g_entry_bits) | ((g_det_bits > 0xffff ? 0xffffu : g_det_bits) << 16);
#endif
    fwrite(&rh, sizeof rh, 1, out);
    if (txtlen) fwrite(txt, 1, txtlen, out);
    free(txt);
    free(buf);
    /* the abort path may have left the arenas above the marks; both resets
     * are idempotent (tab.c:106-156, sol.c:74-87). */
    sol_reset_xx(p);
    tab_reset_xx(hq);
  }
  oh.solve_seconds = t_solve;
  oh.total_pivots = total_pivots;
  fseek(out, 0, SEEK_SET);
  fwrite(&oh, sizeof oh, 1, out);
  fclose(out);
  fclose(in);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 3 && !strcmp(argv[1], "dat")) {
    int simplify = 0, a = 2;
    if (!strcmp(argv[a], "-d")) {
      deepest_cut_xx = 1;
      a++;
    }
    if (!strcmp(argv[a], "-z")) {
      simplify = 1;
      a++;
    }
    return mode_dat(argv[a], simplify);
  }
  if (argc >= 2 && !strcmp(argv[1], "pip")) return mode_pip();
  if (argc >= 4 && !strcmp(argv[1], "batch")) return mode_batch(argv[2], argv[3]);
  fprintf(stderr, "usage: refpip dat [-d] [-z] in.dat | pip < in.pip | batch in.bin out.bin\n");
  return 64;
}
