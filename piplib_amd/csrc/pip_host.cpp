// piplib_amd/csrc/pip_host.cpp -- host side of the C ABI (include/piplib_amd.h), layers 1-2.
//
// Owns no algorithmic work: shapes, workspace layout, launches, timing.  The
// decision-tree host (layer 3) lives in pip_tree.cpp.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pip_host.h"

static thread_local char g_err[512];
void pipamd_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
extern "C" const char *pipamd_last_error(void) { return g_err; }
extern "C" int pipamd_version(void) { return PIPAMD_VERSION; }

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
  e->iter_limit = PIPAMD_DETLOG;
  {
    const char *nl = getenv("PIPAMD_NO_LEAN");  // measurement aid, like pipamd_debug_lean(e, 0)
    e->no_lean = nl && nl[0] == '1';
    const char *n2 = getenv("PIPAMD_NO_LEAN2");  // measurement aid: the general one-wave kernel as the second bulk launch
    e->no_lean2 = n2 && n2[0] == '1';
  }
  pthread_mutex_init(&e->dt_lock, nullptr);
  *out = e;
  return PIPAMD_OK;
}

extern "C" void pipamd_engine_destroy(pipamd_engine *e) {
  if (!e) return;
  for (int i = 0; i < 2 * e->nev; i++) hipEventDestroy(e->ev[i]);
  if (e->d_q) hipFree(e->d_q);
  if (e->h_run) hipHostFree(e->h_run);
  if (e->d_scratch) hipFree(e->d_scratch);
  for (void *b : e->d_side)
    if (b) hipFree(b);
  for (int i = 0; i < e->nretired; i++) hipFree(e->retired[i]);
  if (e->d_side_count) hipFree(e->d_side_count);
  free(e->run);
  for (void *b : e->dt_buf)
    if (b) hipFree(b);
  if (e->dt_host) hipHostFree(e->dt_host);
  pthread_mutex_destroy(&e->dt_lock);
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
  const int ebits = d->entier_bits == 128 ? 128 : 64;
  if (d->entier_bits != 0 && d->entier_bits != 64 && d->entier_bits != 128) {
    pipamd_set_error("entier_bits must be 0 (= 64), 64 or 128");
    return PIPAMD_E_INVALID;
  }
  const int ew = ebits / 64;
  lay->ebits = ebits;
  lay->batch = d->batch;
  lay->nvar = d->nvar;
  lay->nparm = d->nparm;
  lay->ni = d->ni;
  lay->bigparm = d->bigparm;
  lay->tflags = d->tflags & ~(PIPAMD_T_ROWS_STAY | PIPAMD_T_FRESHROWS);  // the load sets FRESHROWS itself
  lay->pad = 0;
  lay->S = d->ni + d->cap_cuts;
  lay->L = round_even(d->nvar + lay->S);
  lay->W = ebits == 128 ? ncol + d->cap_newparm : round_even(ncol + d->cap_newparm);
  // 64-bit tableaux whose row tables outgrow LDS run with the tables in HBM; 128-bit ones must fit
  if (lay->L > PIPAMD_LMAX || lay->S > PIPAMD_SMAX || lay->W > PIPAMD_MAXCOL || lay->S < 1 ||
      (ebits == 128 && pipk_advance_lds_bytes((lay->L + 3) & ~3, (lay->S + 3) & ~3, lay->W, ebits) > PIPAMD_LDS_BUDGET)) {
    pipamd_set_error("batch shape exceeds engine limits (L=%d<=%d, S=%d<=%d, W=%d<=%d, LDS image %zu<=%d bytes)", lay->L,
                     PIPAMD_LMAX, lay->S, PIPAMD_SMAX, lay->W, PIPAMD_MAXCOL,
                     pipk_advance_lds_bytes((lay->L + 3) & ~3, (lay->S + 3) & ~3, lay->W, ebits), PIPAMD_LDS_BUDGET);
    return PIPAMD_E_TOOLARGE;
  }
  if (d->bigparm >= ncol || (d->bigparm >= 0 && d->bigparm <= d->nvar)) {
    pipamd_set_error("bigparm must be -1 or a parameter column (nvar < bigparm < ncol)");
    return PIPAMD_E_INVALID;
  }
  const int64_t sol = round_even((d->nvar * (d->nparm + d->cap_newparm + 1) + d->nvar) * ew);
  const int wp = ebits == 128 ? (lay->W <= 64 ? 64 : (lay->W <= 128 ? 128 : (lay->W <= 256 ? 256 : 512)))
                              : (lay->W <= 128 ? 128 : (lay->W <= 256 ? 256 : 512));
  const int nm = wp / 64;
  // saved summaries, then the determinant log of the last launch (64-bit jobs)
  const int64_t state = round_even(lay->S * nm + (3 * lay->L + 7) / 8) + 2 * PIPAMD_DETLOG * ew;
  lay->sol_words = (int32_t)sol;
  lay->state_words = (int32_t)state;
  // rows: den[L] (entry type) | flag[L] | ref[L];  then S x W entries;  solution;  saved summaries
  lay->per_job = ((int64_t)lay->L * ew + lay->L) + (int64_t)lay->S * lay->W * ew + sol + state;
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

extern "C" size_t pipamd_dense_pivot_bytes(const pipamd_batch_desc *d) {
  // one pivot reads and writes every real row once: 2 * ni * ncol * sizeof(Entier)
  return 2ull * (size_t)d->ni * (size_t)(d->nvar + d->nparm + 1) * (d->entier_bits == 128 ? 16 : 8);
}

extern "C" int pipamd_batch_load_part(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, const int64_t *d_rows,
                                      int first, int count, void *stream) {
  if (e && hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;  // HIP's current device is per host thread
  PipBatchLayout lay;
  size_t jb;
  if (!e || !d_ws || !d_rows) return PIPAMD_E_INVALID;
  int rc = pipamd_batch_layout(d, &lay, &jb);
  if (rc) return rc;
  if (first < 0 || count < 0 || first > lay.batch - count) {
    pipamd_set_error("batch_load_part: tableaux %d..%d outside the batch of %d", first, first + count, lay.batch);
    return PIPAMD_E_INVALID;
  }
  PipJob *jobs = (PipJob *)d_ws;
  long long *arena = (long long *)((char *)d_ws + jb);
  // PIPAMD_T_ROWS_STAY: no copy pass, the first pivot launch reads the caller's rows (whole 16-byte units)
  const int ncol = d->nvar + d->nparm + 1;
  lay.pad = (d->tflags & PIPAMD_T_ROWS_STAY) && lay.ebits != 128 && ncol % 2 == 0 && ((uintptr_t)d_rows & 15) == 0;
  HIPCHK(pipk_launch_batch_load(jobs, arena, (const long long *)d_rows, lay, first, count, (hipStream_t)stream));
  return PIPAMD_OK;
}

extern "C" int pipamd_batch_load(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, const int64_t *d_rows,
                                 void *stream) {
  if (!d) return PIPAMD_E_INVALID;
  return pipamd_batch_load_part(e, d_ws, d, d_rows, 0, d->batch, stream);
}

// traiter() for the whole batch, as a short sequence of launches without a host round trip in
// between (each launch writes the list of tableaux it left unfinished, the next one reads it):
//   bulk  (batches of >= `bulk_min` tableaux, default 2048): one wave per tableau; a tableau's workgroup ends when the
//         tableau is finished, has spent `round_pivots` pivots or has filled the LDS image (sized
//         for the rows it has plus `round_rows` Gomory cuts, so that 24 tableaux fit a CU).  The
//         workgroup dispatcher starts the next tableau in its place, so every CU stays busy until
//         all have had their turn.
//   tail  : what the bulk launch left unfinished (the few tableaux that need many more pivots or
//         rows) runs to completion with four waves per tableau and an image for every spare row:
//         little parallelism is left, so the latency of a pivot is what counts.
// Only then the host looks at the number of tableaux still running (normally 0; tableaux that
// hit the per-launch pivot limit `iter_limit` go through further tail launches).
#define Q_CTRL 3 /* control words per launch: out_count, out_maxni, the number of listed jobs that are out of rows */
// Control words (out_count, out_maxni per launch) must be zero when a launch starts.  They come
// from a pool of Q_POOL solves x Q_FAST launches that one memset zeroes every Q_POOL solves (a
// solve of a small batch is a handful of runtime calls: this was a fifth of them); the rare
// launches beyond Q_FAST of one solve use an overflow area zeroed on demand.
enum { Q_POOL = 64, Q_FAST = 8, Q_NPOOL = Q_CTRL * Q_FAST * Q_POOL, Q_NCTRL = Q_NPOOL + Q_CTRL * PIPAMD_MAX_ROUNDS };

// One pipamd_batch_solve in progress: everything the launch sequence needs between its asynchronous first part
// (bulk + first tail launch + the copy of the tail's control words, pipamd_batch_solve_async) and its completion
// (pipamd_batch_wait: further tail launches / re-housing while tableaux are left).
struct BatchRun {
  pipamd_engine *e;
  pipamd_batch_desc d;
  PipBatchLayout lay;
  PipJob *jobs;
  long long *arena;
  hipStream_t st;
  int *pool, *over, *list[2];
  bool over_zeroed, have_list, active, finishing, replay_pending;
  int stage;       // launches issued
  int curS;        // row capacity of the largest block in play (grows when tableaux are re-housed)
  int grow_round;
  int upper;       // what the host knows about the length of the next input list
  int tail_waves;

  int *ctrl_of(int stg) const { return stg < Q_FAST ? pool + Q_CTRL * stg : over + Q_CTRL * (stg - Q_FAST); }

  int next_stage() {
    if (stage >= PIPAMD_MAX_ROUNDS) {
      pipamd_set_error("batch_solve: more than %d launches", PIPAMD_MAX_ROUNDS);
      return PIPAMD_E_SOLVER;
    }
    if (stage >= Q_FAST && !over_zeroed) {
      HIPCHK(hipMemsetAsync(over, 0, (size_t)Q_CTRL * PIPAMD_MAX_ROUNDS * sizeof(int), st));
      over_zeroed = true;
    }
    return PIPAMD_OK;
  }

  // expanser (traiter.c:55-88, called by integrer when the tableau is full, integrer.c:410-415) for the `n` jobs of
  // the last launch's out list: those at PIPAMD_ST_CAPACITY move into blocks of twice the row capacity in a side arena
  // of the engine and go on; the others are passed through.  A stage like a launch: it consumes the list and writes
  // the next one.  When the engine's limits (16-bit row codes; 128-bit entries: the LDS image) allow no larger
  // block, the pass only drops the jobs that are at the limit from the list: they keep PIPAMD_ST_CAPACITY.
  int rehouse(int n, int ncap) {
    pipamd_batch_desc d2 = d;
    int newS = curS * 2 > curS + 32 ? curS * 2 : curS + 32;
    if (e->grow_step > 0) newS = curS + e->grow_step;  // testing aid: many small growth rounds
    // the caller's row budget (pipamd_engine_set_max_rows); never below what the batch was loaded with
    const int budget = e->max_rows > 0 ? (e->max_rows > lay.S ? e->max_rows : lay.S) : PIPAMD_SMAX;
    if (newS > budget) newS = budget;
    if (newS < curS) newS = curS;  // == curS: nothing larger is allowed -- the pass drops the jobs that are at the limit
    PipBatchLayout nl;
    size_t jb2;
    // the side arena holds the jobs that are out of rows (`ncap`, counted by the launch that listed them), not the whole
    // list: that also carries the jobs merely paused
    if (ncap < 1) ncap = 1;
    if (ncap > n) ncap = n;
    d2.batch = ncap;
    for (;;) {
      d2.cap_cuts = newS - d.ni;
      if (pipamd_batch_layout(&d2, &nl, &jb2) == PIPAMD_OK) break;
      if (newS <= curS) return PIPAMD_E_SOLVER;  // the shape the batch was loaded with (or grew to): cannot fail
      newS = curS + (newS - curS) / 2;
    }
    int rc2 = next_stage();
    if (rc2) return rc2;
    const bool room = grow_round < PIPAMD_MAX_GROW;
    const size_t per_job_bytes = (size_t)nl.per_job * sizeof(int64_t);
    const size_t need = per_job_bytes * (size_t)ncap + 16;
    int side_cap = 0;
    if (room) {
      if (e->side_bytes[grow_round] < need) {
        // grown geometrically; the outgrown arena is kept until the engine goes (hipFree in mid-stream would wait for
        // every other batch in flight on the device); if the device has no room the jobs that do not fit the arena that
        // is there keep PIPAMD_ST_CAPACITY -- the batch is not failed for them
        size_t want = 2 * e->side_bytes[grow_round] > need ? 2 * e->side_bytes[grow_round] : need;
        void *nb = nullptr;
        if (hipMalloc(&nb, want) != hipSuccess) {
          (void)hipGetLastError();
          want = need;
          if (hipMalloc(&nb, want) != hipSuccess) {
            (void)hipGetLastError();
            nb = nullptr;
          }
        }
        if (nb) {
          if (e->d_side[grow_round]) {
            if (e->nretired < (int)(sizeof e->retired / sizeof e->retired[0]))
              e->retired[e->nretired++] = e->d_side[grow_round];
            else
              HIPCHK(hipFree(e->d_side[grow_round]));
          }
          e->d_side[grow_round] = nb;
          e->side_bytes[grow_round] = want;
        }
      }
      const size_t fit = e->side_bytes[grow_round] > 16 ? (e->side_bytes[grow_round] - 16) / per_job_bytes : 0;
      side_cap = fit < (size_t)ncap ? (int)fit : ncap;
    }
    if (!e->d_side_count) HIPCHK(hipMalloc((void **)&e->d_side_count, sizeof(int)));
    HIPCHK(hipMemsetAsync(e->d_side_count, 0, sizeof(int), st));
    // job offsets are in int64 units from `arena`, rows 16-byte aligned
    const char *side = (room && e->d_side[grow_round]) ? (const char *)e->d_side[grow_round] : (const char *)arena;
    if (((side - (const char *)arena) & 15) != 0) side += 8;
    nl.arena_off = (int64_t)((side - (const char *)arena) / (ptrdiff_t)sizeof(int64_t));
    int *c = ctrl_of(stage);
    void *q5[5] = {list[(stage - 1) & 1], ctrl_of(stage - 1), list[stage & 1], c, c + 1};
    HIPCHK(pipk_launch_rehouse(jobs, arena, q5, n, nl, e->d_side_count, side_cap, st));
    stage++;
    if (room) grow_round++;
    curS = newS;
    e->last_rehoused += ncap;
    return PIPAMD_OK;
  }

  int launch(int waves, int budget, int smax, int grid, bool lean = false, bool replay = true) {
    int rcs = next_stage();
    if (rcs) return rcs;
    if (smax > curS) smax = curS;
    int *c = ctrl_of(stage);
    void *q5[5] = {nullptr, nullptr, list[stage & 1], c, c + 1};
    if (have_list) {
      q5[0] = list[(stage - 1) & 1];
      q5[1] = ctrl_of(stage - 1);
    }
    if (!e->no_timing) {
      if (e->nlaunch >= e->nev) {
        HIPCHK(hipEventCreate(&e->ev[2 * e->nev]));
        HIPCHK(hipEventCreate(&e->ev[2 * e->nev + 1]));
        e->nev++;
      }
      HIPCHK(hipEventRecord(e->ev[2 * e->nlaunch], st));
    }
    void *big[2] = {&e->d_scratch, &e->scratch_bytes};
    // a uniform batch without parameters whose rows fill a wave's 128 columns exactly: FULL kernels
    const int hints = ((lay.nparm == 0 && lay.bigparm < 0 && lay.nvar + 1 == 128 && lay.W == 128) ? 1 : 0) |
                      (lean ? (lay.ebits == 128 ? 8 : 2) : 0) | (replay ? 0 : 4);
    HIPCHK(pipk_launch_advance_q(jobs, arena, lay.batch, lay.nvar + smax, smax, lay.W, budget, waves, lay.ebits, q5, grid,
                                 big, hints, e->d_prof, st));
    if (!e->no_timing) HIPCHK(hipEventRecord(e->ev[2 * e->nlaunch + 1], st));
    e->nlaunch++;
    stage++;
    have_list = true;
    return PIPAMD_OK;
  }

  // a tail launch over what is left, and the copy of its control words to the host
  int tail() {
    // The determinant logs of the bulk launches are replayed once, behind the first tail launch, for all tableaux
    // (pipk_launch_replay_all): a replay kernel between two launches is a tiny launch that waits milliseconds for a
    // turn on a device busy with other batches' bulk launches, and nothing in the pivot launches depends on it -- a
    // tableau that overflowed goes on pivoting until the replay says so, its status and pivot count are the same.
    // The bulk launches log at most 2 x 96 pivots per tableau, the log holds PIPAMD_DETLOG.
    // (128-bit entries: a list shorter than the GPU has CUs gets a CU per tableau -- sixteen waves; what is left after the
    // first tail launch are the few tableaux of hundreds of pivots and rows, each a latency chain of its own)
    const bool wide16 = lay.ebits == 128 && lay.W <= 256 && upper <= 256 && !e->waves_per_job && !e->tail_waves;
    int rc = launch(wide16 ? 16 : tail_waves, e->iter_limit, curS, upper, false, !replay_pending);
    if (rc) return rc;
    if (replay_pending) {
      HIPCHK(pipk_launch_replay_all(jobs, arena, lay.batch, lay.ebits, e->lone_batches, st));
      replay_pending = false;
    }
    HIPCHK(hipMemcpyAsync(e->h_run, ctrl_of(stage - 1), Q_CTRL * sizeof(int), hipMemcpyDeviceToHost, st));
    return PIPAMD_OK;
  }

  int begin(pipamd_engine *e_, void *d_ws, const pipamd_batch_desc *d_, void *stream) {
    e = e_;
    size_t jb;
    int rc = pipamd_batch_layout(d_, &lay, &jb);
    if (rc) return rc;
    d = *d_;
    jobs = (PipJob *)d_ws;
    arena = (long long *)((char *)d_ws + jb);
    st = (hipStream_t)stream;
    if (!e->h_run) HIPCHK(hipHostMalloc((void **)&e->h_run, Q_CTRL * sizeof(int), hipHostMallocDefault));
    if (!e->d_q || e->q_cap < lay.batch) {
      if (e->d_q) HIPCHK(hipFree(e->d_q));
      e->d_q = nullptr;
      HIPCHK(hipMalloc((void **)&e->d_q, ((size_t)Q_NCTRL + 2 * (size_t)lay.batch) * sizeof(int)));
      e->q_cap = lay.batch;
      e->solve_seq = 0;
    }
    if (e->solve_seq % Q_POOL == 0 || e->pool_stream != st) {
      HIPCHK(hipMemsetAsync(e->d_q, 0, (size_t)Q_NPOOL * sizeof(int), st));
      e->solve_seq = 0;
      e->pool_stream = st;
    }
    pool = e->d_q + Q_CTRL * Q_FAST * (e->solve_seq % Q_POOL);
    over = e->d_q + Q_NPOOL;
    e->solve_seq++;
    over_zeroed = false;
    list[0] = e->d_q + Q_NCTRL;
    list[1] = e->d_q + Q_NCTRL + e->q_cap;
    e->nlaunch = 0;
    e->last_rehoused = 0;
    stage = 0;
    have_list = false;
    finishing = false;
    replay_pending = false;
    curS = lay.S;
    grow_round = 0;
    const bool integer = (lay.tflags & PIPAMD_T_INT) != 0;
    const int K1 = e->round_pivots > 0 ? e->round_pivots : 96;
    const int KA = !integer ? 0 : (e->round_rows > 0 ? e->round_rows : 48);
    tail_waves = e->waves_per_job ? e->waves_per_job : (e->tail_waves ? e->tail_waves : 4);
    upper = lay.batch;
    // The 128-bit flavour on rows of 129 ... 256 columns without parameters can start with the lean kernel of pip_lean64.h,
    // one wave per tableau over the whole batch: it finishes the tableaux whose stored entries stay below 2^63 (two in
    // three of BASELINE's configs[4]) and leaves the others to the launches below.  Opt-in (pipamd_engine_set_lean64):
    // measured on that batch it is no faster than pip_advance_kernel's four waves per tableau.
    bool took64 = false;
    if (e->lean64 && !e->no_lean && lay.ebits == 128 && lay.nparm == 0 && lay.bigparm < 0 && lay.W > 128 && lay.W <= 256 &&
        !(lay.tflags & (PIPAMD_T_NOSKIP | PIPAMD_T_DEEPEST)) && lay.batch >= (e->bulk_min > 0 ? e->bulk_min : 128) &&
        e->waves_per_job != 4 && e->waves_per_job != 8) {
      int smax = curS < 448 ? curS : 448;  // (448 rows: an image of 30 KB, five tableaux per CU)
      if (smax >= lay.ni && pipk_lean64_lds_bytes(smax, lay.nvar + smax) <= PIPAMD_LDS_BUDGET) {
        rc = launch(1, e->iter_limit, smax, lay.batch, true, true);
        if (rc) return rc;
        took64 = true;
        if (e->single_launch) {  // measurement aid: the lean launch on its own (its unfinished tableaux stay PIPAMD_ST_RUN)
          HIPCHK(hipMemcpyAsync(e->h_run, ctrl_of(stage - 1), Q_CTRL * sizeof(int), hipMemcpyDeviceToHost, st));
          active = true;
          return PIPAMD_OK;
        }
      }
    }
    if (!took64 && lay.batch >= (e->bulk_min > 0 ? e->bulk_min : 2048) && e->waves_per_job != 4 && e->waves_per_job != 8) {
      const int budget = e->iter_limit < K1 ? e->iter_limit : K1;
      int smax = lay.ni + (KA < budget ? KA : budget);
      if (smax > curS) smax = curS;
      // no parameters, at most 128 columns of 64-bit entries, rows skipped, plain cuts: the lean kernel (pip_lean.h) goes
      // first -- it finishes the tableaux whose entries stay below 2^15 and leaves the others to the general kernel's launch
      const bool lean = !e->no_lean && lay.ebits != 128 && lay.nparm == 0 && lay.bigparm < 0 && lay.W <= 128 && !(lay.W & 1) &&
                        !(lay.tflags & (PIPAMD_T_NOSKIP | PIPAMD_T_DEEPEST)) && pipk_lean_class(smax) != 0;
      // (the bulk launches leave their determinant logs to the replay behind the first tail launch; two of them log at
      // most 2 * budget pivots per tableau)
      const bool defer = 2 * budget <= PIPAMD_DETLOG / 2;
      if (lean) {
        rc = launch(1, budget, smax, lay.batch, true, !defer);
        if (rc) return rc;
        replay_pending = defer;
        if (e->single_launch == 2) {  // measurement aid: the lean launch on its own
          if (replay_pending) HIPCHK(pipk_launch_replay_all(jobs, arena, lay.batch, lay.ebits, e->lone_batches, st));
          replay_pending = false;
          HIPCHK(hipMemcpyAsync(e->h_run, ctrl_of(stage - 1), Q_CTRL * sizeof(int), hipMemcpyDeviceToHost, st));
          active = true;
          return PIPAMD_OK;
        }
      }
      // Then what the lean launch left (on the headline: the 6 % of the tableaux that spent its pivot budget) in a second
      // one-wave launch: the lean kernel again, which resumes its own paused jobs, with the largest row-capacity class and
      // a budget that covers the longest tableaux -- and leaves what it cannot take (rows beyond ints, more rows than its
      // image) to the tail.  Where the shape has no lean kernel: the general one-wave kernel.  (Measured with 14 batches
      // in flight: sending those tableaux straight to the four-wave tail instead costs 10 % of the throughput.)
      if (!(lean && e->lone_batches)) {  // (pipamd_engine_set_lone_batches: straight to the tail launches)
        const bool lean2 = lean && !e->no_lean2 && pipk_lean_class(lay.ni + 96 < curS ? lay.ni + 96 : curS) != 0;
        if (lean2) {
          int smax2 = lay.ni + 96 < curS ? lay.ni + 96 : curS;
          const int b2 = e->iter_limit < 160 ? e->iter_limit : 160;
          rc = launch(1, b2, smax2, lay.batch, true, false);
          replay_pending = true;
        } else {
          rc = launch(1, budget, smax, lay.batch, false, !defer);
          replay_pending = defer;
        }
        if (rc) return rc;
      }
      if (e->single_launch) {  // measurement aid: the bulk launch on its own (its tableaux stay PIPAMD_ST_RUN)
        if (replay_pending) HIPCHK(pipk_launch_replay_all(jobs, arena, lay.batch, lay.ebits, e->lone_batches, st));
        replay_pending = false;
        HIPCHK(hipMemcpyAsync(e->h_run, ctrl_of(stage - 1), Q_CTRL * sizeof(int), hipMemcpyDeviceToHost, st));
        active = true;
        return PIPAMD_OK;
      }
    }
    rc = tail();
    if (rc) return rc;
    active = true;
    return PIPAMD_OK;
  }

  int wait_stream() {
    if (e->blocking_wait) {  // sleep between looks at the stream instead of spinning on it (pipamd_engine_set_blocking_wait)
      hipError_t q;
      const struct timespec nap = {0, 40000};
      while ((q = hipStreamQuery(st)) == hipErrorNotReady) nanosleep(&nap, nullptr);
      HIPCHK(q);
    } else {
      HIPCHK(hipStreamSynchronize(st));
    }
    return PIPAMD_OK;
  }

  // What the launches enqueued so far left behind (the stream must be idle): 1 = every tableau has its final status,
  // 0 = more work was enqueued (a further tail launch, after re-housing the tableaux that are out of rows), < 0 error.
  int advance() {
    if (finishing) {  // the solutions of re-housed tableaux are back in the caller's workspace
      finishing = false;
      active = false;
      e->timed = !e->no_timing;
      return 1;
    }
    if (e->h_run[0] > 0 && !e->single_launch) {
      upper = e->h_run[0];
      if (e->h_run[1] & PIPAMD_Q_CAPFLAG) {  // some of them have spent their spare rows
        int rc = rehouse(upper, e->h_run[2]);
        if (rc) return rc;
      }
      int rc = tail();
      return rc ? rc : 0;
    }
    if (grow_round > 0) {  // the side arenas serve the engine's next solve: the solve ends when the copy-back has
      HIPCHK(pipk_launch_rehouse_finish(jobs, arena, lay.batch, lay.sol_words, st));
      finishing = true;
      return 0;
    }
    active = false;
    e->timed = !e->no_timing;
    return 1;
  }

  int finish() {
    for (;;) {
      int rc = wait_stream();
      if (rc) {
        active = false;
        return rc;
      }
      rc = advance();
      if (rc < 0) active = false;
      if (rc != 0) return rc < 0 ? rc : PIPAMD_OK;
    }
  }
};

static int batch_begin(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, void *stream) {
  if (e && hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;  // HIP's current device is per host thread
  if (!e || !d_ws || !d) return PIPAMD_E_INVALID;
  if (!e->run) {
    e->run = (BatchRun *)calloc(1, sizeof(BatchRun));
    if (!e->run) return PIPAMD_E_NOMEM;
  }
  if (e->run->active) {
    pipamd_set_error("this engine already has a batch solve in flight: pipamd_batch_wait first (one per engine; engines are cheap)");
    return PIPAMD_E_INVALID;
  }
  return e->run->begin(e, d_ws, d, stream);
}

extern "C" int pipamd_batch_solve(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, void *stream) {
  int rc = batch_begin(e, d_ws, d, stream);
  if (rc) {
    if (e && e->run) e->run->active = false;
    return rc;
  }
  return e->run->finish();
}

extern "C" int pipamd_batch_solve_async(pipamd_engine *e, void *d_ws, const pipamd_batch_desc *d, void *stream) {
  int rc = batch_begin(e, d_ws, d, stream);
  if (rc && e && e->run) e->run->active = false;
  return rc;
}

extern "C" int pipamd_batch_wait(pipamd_engine *e) {
  if (e && hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;
  if (!e) return PIPAMD_E_INVALID;
  if (!e->run || !e->run->active) return PIPAMD_OK;  // nothing in flight
  return e->run->finish();
}

extern "C" int pipamd_batch_poll(pipamd_engine *e) {
  if (!e) return PIPAMD_E_INVALID;
  if (!e->run || !e->run->active) return 1;
  if (hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;
  hipError_t q = hipStreamQuery(e->run->st);
  if (q == hipErrorNotReady) return 0;
  if (q != hipSuccess) {
    pipamd_set_error("hipStreamQuery failed: %s", hipGetErrorString(q));
    e->run->active = false;
    return PIPAMD_E_HIP;
  }
  // the launches enqueued so far have ended: done, or the next ones go out now (never blocks)
  int rc = e->run->advance();
  if (rc < 0) e->run->active = false;
  return rc;
}

// Row budget of pipamd_batch_solve: a tableau is re-housed (expanser) while its row capacity stays within `rows`
// (0 = the default: only the engine's own limit); beyond it the tableau ends PIPAMD_ST_CAPACITY.  The reference grows
// without bound, and so does the default here; a caller that feeds tableaux on which Gomory cuts converge slowly
// (thousands of cut rows, each pivot then rewriting thousands of rows) bounds the memory and time of a batch with it.
extern "C" int pipamd_engine_set_max_rows(pipamd_engine *e, int rows) {
  if (!e || rows < 0) return PIPAMD_E_INVALID;
  e->max_rows = rows;
  return PIPAMD_OK;
}

// Testing aid: a tableau that has spent its spare rows is re-housed with `rows` more instead of twice as many, so
// that a test sees many growth rounds on small inputs (0 = the default doubling).
extern "C" int pipamd_debug_grow_step(pipamd_engine *e, int rows) {
  if (!e || rows < 0) return PIPAMD_E_INVALID;
  e->grow_step = rows;
  return PIPAMD_OK;
}

// Measurement aid (bench.py's per-launch roofline): the solve stops after its first launch -- the one-wave bulk
// launch when the batch has one -- leaving unfinished tableaux at PIPAMD_ST_RUN; pipamd_batch_counters then tells
// what that launch alone did.
extern "C" int pipamd_debug_single_launch(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->single_launch = on < 0 ? 0 : (on > 2 ? 1 : on);  // 1: after the bulk launches, 2: after the lean launch alone
  return PIPAMD_OK;
}

// Measurement / testing aid: without the lean kernel every tableau of a bulk launch runs in pip_advance_kernel.
extern "C" int pipamd_debug_lean(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->no_lean = on ? 0 : 1;
  return PIPAMD_OK;
}

// The lean kernel of the 128-bit flavour (csrc/pip_lean64.h) as the first launch of pipamd_batch_solve on batches it can
// take (no parameters, 129 ... 256 columns, at least 128 tableaux); default off.
extern "C" int pipamd_engine_set_lean64(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->lean64 = on ? 1 : 0;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_timing(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->no_timing = !on;
  return PIPAMD_OK;
}

// Sum of the advance-kernel launch durations of the last pipamd_batch_solve (HIP events on
// the launch stream) and their number.
extern "C" int pipamd_last_solve_ms(pipamd_engine *e, float *ms) {
  if (!e || !ms || !e->timed) return PIPAMD_E_INVALID;
  float tot = 0;
  for (int i = 0; i < e->nlaunch; i++) {
    float t = 0;
    HIPCHK(hipEventSynchronize(e->ev[2 * i + 1]));
    HIPCHK(hipEventElapsedTime(&t, e->ev[2 * i], e->ev[2 * i + 1]));
    tot += t;
  }
  *ms = tot;
  return PIPAMD_OK;
}

extern "C" int pipamd_last_solve_launches(pipamd_engine *e) { return e ? e->nlaunch : 0; }

// Duration of launch `i` of the last pipamd_batch_solve (diagnostics, tools/).
extern "C" int pipamd_last_launch_ms(pipamd_engine *e, int i, float *ms) {
  if (!e || !ms || !e->timed || i < 0 || i >= e->nlaunch) return PIPAMD_E_INVALID;
  HIPCHK(hipEventSynchronize(e->ev[2 * i + 1]));
  HIPCHK(hipEventElapsedTime(ms, e->ev[2 * i], e->ev[2 * i + 1]));
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_round_pivots(pipamd_engine *e, int pivots) {
  if (!e || pivots < 1) return PIPAMD_E_INVALID;
  e->round_pivots = pivots;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_tail_waves(pipamd_engine *e, int waves) {
  if (!e || (waves != 4 && waves != 8)) return PIPAMD_E_INVALID;
  e->tail_waves = waves;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_lone_batches(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->lone_batches = on ? 1 : 0;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_blocking_wait(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->blocking_wait = on ? 1 : 0;
  return PIPAMD_OK;
}
extern "C" int pipamd_engine_set_device_tree(pipamd_engine *e, int on) {
  if (!e) return PIPAMD_E_INVALID;
  e->no_device_tree = on ? 0 : 1;
  return PIPAMD_OK;
}
extern "C" int pipamd_last_device_tree(const pipamd_engine *e, int *served, int *handed_back) {
  if (!e) return PIPAMD_E_INVALID;
  if (served) *served = e->dt_served;
  if (handed_back) *handed_back = e->dt_fallback;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_bulk_min(pipamd_engine *e, int tableaux) {
  if (!e || tableaux < 1) return PIPAMD_E_INVALID;
  e->bulk_min = tableaux;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_round_rows(pipamd_engine *e, int rows) {
  if (!e || rows < 1) return PIPAMD_E_INVALID;
  e->round_rows = rows;
  return PIPAMD_OK;
}

extern "C" int pipamd_batch_results(pipamd_engine *e, const void *d_ws, const pipamd_batch_desc *d, int32_t *d_status,
                                    int32_t *d_pivots, int32_t *d_cuts, int64_t *d_sol_num, int64_t *d_sol_den,
                                    void *stream) {
  if (e && hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;  // HIP's current device is per host thread
  PipBatchLayout lay;
  size_t jb;
  if (!e || !d_ws) return PIPAMD_E_INVALID;
  int rc = pipamd_batch_layout(d, &lay, &jb);
  if (rc) return rc;
  const PipJob *jobs = (const PipJob *)d_ws;
  const long long *arena = (const long long *)((const char *)d_ws + jb);
  HIPCHK(pipk_launch_batch_results(jobs, arena, lay.batch, lay.nvar, lay.nparm, lay.ebits, d_status, d_pivots, d_cuts,
                                   (void *)d_sol_num, (void *)d_sol_den, (hipStream_t)stream));
  return PIPAMD_OK;
}

extern "C" int pipamd_batch_counters(pipamd_engine *e, const void *d_ws, const pipamd_batch_desc *d, uint64_t *d_out4,
                                     void *stream) {
  if (e && hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;  // HIP's current device is per host thread
  PipBatchLayout lay;
  size_t jb;
  if (!e || !d_ws || !d_out4) return PIPAMD_E_INVALID;
  int rc = pipamd_batch_layout(d, &lay, &jb);
  if (rc) return rc;
  HIPCHK(pipk_launch_batch_counters((const PipJob *)d_ws, lay.batch, (unsigned long long *)d_out4, (hipStream_t)stream));
  return PIPAMD_OK;
}

// Diagnostic only: per-phase cycle sums of the advance kernel (needs a -DPIP_PROFILE build).
extern "C" int pipamd_debug_profile(pipamd_engine *e, int enable, uint64_t *host_out10) {
  if (!e) return PIPAMD_E_INVALID;
  if (enable && !e->d_prof) {
    HIPCHK(hipMalloc((void **)&e->d_prof, 64 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(e->d_prof, 0, 64 * sizeof(unsigned long long)));
  }
  if (host_out10 && e->d_prof) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(host_out10, e->d_prof, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->d_prof, 0, 64 * sizeof(unsigned long long)));
  }
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_waves_per_job(pipamd_engine *e, int waves) {
  if (!e || (waves != 0 && waves != 1 && waves != 4 && waves != 8)) return PIPAMD_E_INVALID;
  e->waves_per_job = waves;
  return PIPAMD_OK;
}

extern "C" int pipamd_engine_set_iter_limit(pipamd_engine *e, int pivots_per_launch) {
  if (!e || pivots_per_launch < 1) return PIPAMD_E_INVALID;
  // a launch logs (pivot, denominator) per pivot for the determinant replay: PIPAMD_DETLOG entries per job
  e->iter_limit = pivots_per_launch < PIPAMD_DETLOG ? pivots_per_launch : PIPAMD_DETLOG;
  return PIPAMD_OK;
}

extern "C" void pipamd_free(void *p) { free(p); }
