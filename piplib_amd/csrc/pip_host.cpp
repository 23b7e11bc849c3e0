// piplib_amd/csrc/pip_host.cpp -- host side of the C ABI (include/piplib_amd.h), layers 1-2.
//
// Owns no algorithmic work: shapes, workspace layout, launches, timing.  The
// decision-tree host (layer 3) lives in pip_tree.cpp.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pip_host.h"

static thread_local char g_err[512];
void pipamd_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
extern "C" const char *pipamd_last_error(void) { return g_err; }
extern "C" int pipamd_version(void) { return 100; }

#define HIPCHK(call)                                                                    \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      pipamd_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return PIPAMD_E_HIP;                                                              \
    }                                                                                   \
  } while (0)

extern "C" int pipamd_engine_create(pipamd_engine **out, int device) {
  if (!out) return PIPAMD_E_INVALID;
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (n <= 0 || device < 0 || device >= n) {
    pipamd_set_error("no such HIP device %d (count %d): the HIP path is mandatory, there is no CPU fallback", device, n);
    return PIPAMD_E_HIP;
  }
  HIPCHK(hipSetDevice(device));
  pipamd_engine *e = (pipamd_engine *)calloc(1, sizeof *e);
  if (!e) return PIPAMD_E_NOMEM;
  e->device = device;
  e->iter_limit = 1 << 20;
  HIPCHK(hipEventCreate(&e->ev0));
  HIPCHK(hipEventCreate(&e->ev1));
  *out = e;
  return PIPAMD_OK;
}

extern "C" void pipamd_engine_destroy(pipamd_engine *e) {
  if (!e) return;
  hipEventDestroy(e->ev0);
  hipEventDestroy(e->ev1);
  if (e->d_scratch) hipFree(e->d_scratch);
  free(e);
}

// ------------------------------------------------------------------ layer 1
static int round_even(int x) { return (x + 1) & ~1; }

int pipamd_batch_layout(const pipamd_batch_desc *d, PipBatchLayout *lay, size_t *jobs_bytes) {
  if (!d || d->batch <= 0 || d->nvar < 0 || d->nparm < 0 || d->ni < 0 || d->cap_cuts < 0 || d->cap_newparm < 0) {
    pipamd_set_error("invalid batch descriptor");
    return PIPAMD_E_INVALID;
  }
  const int ncol = d->nvar + d->nparm + 1;
  lay->batch = d->batch;
  lay->nvar = d->nvar;
  lay->nparm = d->nparm;
  lay->ni = d->ni;
  lay->bigparm = d->bigparm;
  lay->tflags = d->tflags;
  lay->S = d->ni + d->cap_cuts;
  lay->L = round_even(d->nvar + lay->S);
  lay->W = round_even(ncol + d->cap_newparm);
  if (lay->L > PIPAMD_LMAX || lay->S > PIPAMD_SMAX || lay->W > PIPAMD_MAXCOL || lay->S < 1) {
    pipamd_set_error("batch shape exceeds engine limits (L=%d<=%d, S=%d<=%d, W=%d<=%d)", lay->L, PIPAMD_LMAX, lay->S,
                     PIPAMD_SMAX, lay->W, PIPAMD_MAXCOL);
    return PIPAMD_E_TOOLARGE;
  }
  if (d->bigparm >= ncol || (d->bigparm >= 0 && d->bigparm <= d->nvar)) {
    pipamd_set_error("bigparm must be -1 or a parameter column (nvar < bigparm < ncol)");
    return PIPAMD_E_INVALID;
  }
  const int64_t sol = round_even(d->nvar * (d->nparm + d->cap_newparm + 1) + d->nvar);
  lay->per_job = 2 * (int64_t)lay->L + (int64_t)lay->S * lay->W + sol;
  lay->arena_off = 0;
  *jobs_bytes = ((size_t)d->batch * sizeof(PipJob) + 255) & ~(size_t)255;
  return PIPAMD_OK;
}

extern "C" size_t pipamd_batch_workspace_bytes(const pipamd_batch_desc *d) {
  PipBatchLayout lay;
  size_t jb;
  if (pipamd_batch_layout(d, &lay, &jb) != PIPAMD_OK) return 0;
  return jb + (size_t)lay.per_job * (size_t)d->batch * sizeof(int64_t);
}

extern "C" size_t pipamd_pivot_bytes(const pipamd_batch_desc *d) {
  // one pivot reads and writes every real row once: 2 * ni * ncol * sizeof(Entier)
  return 2ull * (size_t)d->ni * (size_t)(d->nvar + d->nparm + 1) * sizeof(int64_t);
}

extern "C" int pipamd_batch_load(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, const int64_t *d_rows,
                                 void *stream) {
  PipBatchLayout lay;
  size_t jb;
  if (!e || !d_ws || !d_rows) return PIPAMD_E_INVALID;
  int rc = pipamd_batch_layout(d, &lay, &jb);
  if (rc) return rc;
  PipJob *jobs = (PipJob *)d_ws;
  long long *arena = (long long *)((char *)d_ws + jb);
  HIPCHK(pipk_launch_batch_load(jobs, arena, (const long long *)d_rows, lay, (hipStream_t)stream));
  return PIPAMD_OK;
}

extern "C" int pipamd_batch_solve(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, void *stream) {
  PipBatchLayout lay;
  size_t jb;
  if (!e || !d_ws) return PIPAMD_E_INVALID;
  int rc = pipamd_batch_layout(d, &lay, &jb);
  if (rc) return rc;
  PipJob *jobs = (PipJob *)d_ws;
  long long *arena = (long long *)((char *)d_ws + jb);
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(hipEventRecord(e->ev0, st));
  HIPCHK(pipk_launch_advance(jobs, arena, lay.batch, lay.L, lay.S, lay.W, e->iter_limit, st));
  HIPCHK(hipEventRecord(e->ev1, st));
  e->timed = 1;
  return PIPAMD_OK;
}

extern "C" int pipamd_last_solve_ms(pipamd_engine *e, float *ms) {
  if (!e || !ms || !e->timed) return PIPAMD_E_INVALID;
  HIPCHK(hipEventSynchronize(e->ev1));
  HIPCHK(hipEventElapsedTime(ms, e->ev0, e->ev1));
  return PIPAMD_OK;
}

extern "C" int pipamd_batch_results(pipamd_engine *e, const void *d_ws, const pipamd_batch_desc *d, int32_t *d_status,
                                    int32_t *d_pivots, int32_t *d_cuts, int64_t *d_sol_num, int64_t *d_sol_den,
                                    void *stream) {
  PipBatchLayout lay;
  size_t jb;
  if (!e || !d_ws) return PIPAMD_E_INVALID;
  int rc = pipamd_batch_layout(d, &lay, &jb);
  if (rc) return rc;
  const PipJob *jobs = (const PipJob *)d_ws;
  const long long *arena = (const long long *)((const char *)d_ws + jb);
  HIPCHK(pipk_launch_batch_results(jobs, arena, lay.batch, lay.nvar, lay.nparm, d_status, d_pivots, d_cuts,
                                   (long long *)d_sol_num, (long long *)d_sol_den, (hipStream_t)stream));
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_iter_limit(pipamd_engine *e, int pivots_per_launch) {
  if (!e || pivots_per_launch < 1) return PIPAMD_E_INVALID;
  e->iter_limit = pivots_per_launch;
  return PIPAMD_OK;
}

extern "C" void pipamd_free(void *p) { free(p); }
