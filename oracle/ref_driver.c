/* oracle/ref_driver.c -- TEST INFRASTRUCTURE, not product code.
 *
 * A small driver of our own around the *real* reference library
 * (oracle/_ref/libpiplib_ref_dp.so, compiled by oracle/Makefile straight from
 * /root/reference/source/{piplib,traiter,integrer,tab,sol}.c with
 * -DPIPLIB_INT_DP, i.e. the reference's "pip64"/piplib64 build).
 *
 * The reference's own command-line front end (source/maind.c) cannot be built
 * here because it includes a generated "version.h"; this file replaces it with
 * the minimum needed to (1) turn a .dat file into the .ll text the reference
 * test-suite diffs against (test/Makefile.am:62-87), (2) run the .pip examples
 * through pip_solve (example/example.c), and (3) solve a binary batch of
 * tableaux while counting calls to pivoter_dp, for the CPU baseline.
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

extern int verbose_dp;
extern int deepest_cut_dp;

/* ---- pivot counter: interposes the reference's pivoter_dp (traiter.c:345) ----
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
static int (*real_pivoter)(Tableau_dp *, int, int, int, int);
int pivoter_dp(Tableau_dp *tp, int pivi, int nvar, int nparm, int ni) {
  if (!real_pivoter)
    real_pivoter = (int (*)(Tableau_dp *, int, int, int, int))dlsym(RTLD_NEXT, "pivoter_dp");
  g_pivots++;
  return real_pivoter(tp, pivi, nvar, nparm, ni);
}
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
static int run_traiter(Tableau_dp *ineq, Tableau_dp *context, int nvar, int nparm, int ni, int nc,
                       int bigparm, int nq, int p) {
  int non_vide = 1;
  if (nc) {
    Tableau_dp *ctxt = expanser_dp(context, nparm, nc, nparm + 1, nparm, 0, 0);
    traiter_dp(ctxt, NULL, nparm, 0, nc, 0, -1, TRAITER_INT);
    non_vide = is_not_Nil_dp(p);
    sol_reset_dp(p);
  }
  if (non_vide) traiter_dp(ineq, context, nvar, nparm, ni, nc, bigparm, nq ? TRAITER_INT : 0);
  return non_vide;
}

/* ------------------------------------------------------------------ dat mode */
static int read_int(FILE *in, int *v) {
  long long x;
  if (dscanf_dp(in, &x) < 0) return -1;
  *v = (int)x;
  return 0;
}

static int mode_dat(const char *path, int simplify) {
  FILE *in = fopen(path, "r");
  FILE *out = stdout;
  int c;
  if (!in) {
    fprintf(stderr, "%s unaccessible\n", path);
    return 1;
  }
  verbose_dp = -1;
  sol_init_dp();
  tab_init_dp();
  while ((c = dgetc_dp(in)) != EOF) {
    int nvar, nparm, ni, nc, bigparm, nq, level = 0, p, xq, q;
    struct high_water_mark_dp hq;
    Tableau_dp *ineq, *context;
    if (c != '(') continue;
    /* echo the comment group, as the reference front end does */
    fputc('(', out);
    while ((c = dgetc_dp(in)) != EOF) {
      if (c == '(') level++;
      else if (c == ')' && --level == 0) break;
      fputc(c, out);
    }
    if (read_int(in, &nvar) || read_int(in, &nparm) || read_int(in, &ni) || read_int(in, &nc) ||
        read_int(in, &bigparm) || read_int(in, &nq)) {
      fprintf(out, "\nSyntax error\n)\n");
      break;
    }
    hq = tab_hwm_dp();
    ineq = tab_get_dp(in, ni, nvar + nparm + 1, nvar);
    if (!ineq) break;
    if (nq) tab_simplify_dp(ineq, nvar);
    context = tab_get_dp(in, nc, nparm + 1, 0);
    if (!context) break;
    if (nq) tab_simplify_dp(context, nparm);
    xq = p = sol_hwm_dp();
    if (run_traiter(ineq, context, nvar, nparm, ni, nc, bigparm, nq, p)) {
      fputs(")\n", out);
      if (simplify) sol_simplify_dp(xq);
      q = sol_hwm_dp();
      while ((xq = sol_edit_dp(out, xq)) != q)
        ;
      sol_reset_dp(p);
    } else
      fprintf(out, "void\n");
    tab_reset_dp(hq);
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
  PipMatrix_dp *domain, *context;
  PipQuast_dp *solution;
  PipOptions_dp *options;
  printf("[PIP2-like future input] Please enter:\n- the context matrix,\n");
  context = pip_matrix_read_dp(stdin);
  pip_matrix_print_dp(stdout, context);
  printf("- the bignum column (start at 0, -1 if no bignum),\n");
  if (fscanf(stdin, " %d", &bignum) != 1) return 1;
  printf("%d\n", bignum);
  printf("- the constraint matrix.\n");
  domain = pip_matrix_read_dp(stdin);
  pip_matrix_print_dp(stdout, domain);
  printf("\n");
  options = pip_options_init_dp();
  while (fgets(s, sizeof s, stdin)) {
    if (!strncasecmp(s, "Maximize", 8)) options->Maximize = 1;
    if (!strncasecmp(s, "Urs_parms", 9)) options->Urs_parms = 1;
    if (!strncasecmp(s, "Urs_unknowns", 12)) options->Urs_unknowns = 1;
    if (!strncasecmp(s, "Rational", 8)) options->Nq = 0;
    if (!strncasecmp(s, "Dual", 4)) options->Compute_dual = 1;
  }
  if (bignum > 0) bignum += domain->NbColumns - context->NbColumns;
  solution = pip_solve_dp(domain, context, bignum, options);
  pip_quast_print_dp(stdout, solution, 0);
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
  verbose_dp = -1;
  deepest_cut_dp = (bh.flags & BATCH_F_DEEPEST) ? 1 : 0;
  sol_init_dp();
  tab_init_dp();
  for (k = 0; k < bh.count; k++) {
    struct batch_prob ph;
    struct batch_res rh;
    struct high_water_mark_dp hq;
    Tableau_dp *ineq, *context;
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
    hq = tab_hwm_dp();
    xq = p = sol_hwm_dp();
    g_pivots = 0;
    /* tab_get_dp's effect (tab.c:222-248) without the text parsing: rows are
     * Unknown with denominator 1. */
    ineq = tab_alloc_dp(ph.ni, ncol, ph.nvar);
    for (i = 0; i < ph.ni; i++) {
      Flag(ineq, ph.nvar + i) = Unknown;
      Denom(ineq, ph.nvar + i) = 1;
      for (j = 0; j < ncol; j++) Index(ineq, ph.nvar + i, j) = buf[(size_t)i * ncol + j];
    }
    context = tab_alloc_dp(ph.nc, ph.nparm + 1, 0);
    for (i = 0; i < ph.nc; i++) {
      Flag(context, i) = Unknown;
      Denom(context, i) = 1;
      for (j = 0; j <= ph.nparm; j++)
        Index(context, i, j) = buf[(size_t)ph.ni * ncol + (size_t)i * (ph.nparm + 1) + j];
    }
    t0 = now_s();
    g_trap_armed = 1;
    if (setjmp(g_trap) == 0) {
      int nv;
      if (ph.nq && !(bh.flags & BATCH_F_NOSIMPLIFY)) {
        tab_simplify_dp(ineq, ph.nvar);
        tab_simplify_dp(context, ph.nparm);
      }
      nv = run_traiter(ineq, context, ph.nvar, ph.nparm, ph.ni, ph.nc, ph.bigparm, ph.nq, p);
      g_trap_armed = 0;
      t_solve += now_s() - t0;
      rh.status = nv ? BATCH_ST_OK : BATCH_ST_VOID;
      if (nv && !(bh.flags & BATCH_F_NOTEXT)) {
        FILE *ms = open_memstream(&txt, &txtlen);
        q = sol_hwm_dp();
        while ((xq = sol_edit_dp(ms, xq)) != q)
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
    fwrite(&rh, sizeof rh, 1, out);
    if (txtlen) fwrite(txt, 1, txtlen, out);
    free(txt);
    free(buf);
    /* the abort path may have left the arenas above the marks; both resets
     * are idempotent (tab.c:106-156, sol.c:74-87). */
    sol_reset_dp(p);
    tab_reset_dp(hq);
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
      deepest_cut_dp = 1;
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
