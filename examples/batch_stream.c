/* examples/batch_stream.c -- a stream of batches through the asynchronous half of the C ABI, from plain C and ONE host
 * thread: what bench.py's driver does, without Python.
 *
 *   gcc -O2 examples/batch_stream.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lpiplib_amd -lpipamd \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../piplib_amd' -o examples/batch_stream
 *   examples/batch_stream rows.bin <batches> <tableaux> <nvar> <ni> <lanes> <steps>  [<nq> [<lone>]]
 *
 * rows.bin: <batches> x <tableaux> x <ni> x (<nvar>+1) int64, row-major (PIP column order unknowns | constant), no
 * parameters; integer solve (<nq> = 0: rational solve, the reference's Nq).  The batches are made resident in HBM, every batch is solved once for its pivot count,
 * then <steps> steps are timed: step k = pipamd_batch_load + pipamd_batch_solve_async + pipamd_batch_results of batch
 * k mod <batches> on whichever of the <lanes> lanes (engine + workspace + stream) is free; the thread goes round
 * pipamd_batch_poll.  Prints one line: tableaux, pivots, milliseconds, pivots/s, and the status histogram of the last
 * solve of every lane.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime_api.h>

#include "piplib_amd.h"

#define CHECK(x)                                                                                 \
  do {                                                                                           \
    int rc_ = (x);                                                                               \
    if (rc_) {                                                                                   \
      fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, pipamd_last_error());                     \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)
#define HIP(x)                                                                                   \
  do {                                                                                           \
    hipError_t e_ = (x);                                                                         \
    if (e_ != hipSuccess) {                                                                      \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                             \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

typedef struct {
  pipamd_engine *e;
  void *ws;
  hipStream_t st;
  int32_t *status, *pivots, *cuts;
  int64_t *num, *den;
  int busy, batch;
} lane_t;

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int main(int argc, char **argv) {
  if (argc < 8) {
    fprintf(stderr, "usage: %s rows.bin batches tableaux nvar ni lanes steps [nq [lone]]\n", argv[0]);
    return 64;
  }
  const int nb = atoi(argv[2]), B = atoi(argv[3]), nvar = atoi(argv[4]), ni = atoi(argv[5]), K = atoi(argv[6]),
            steps = atoi(argv[7]), nq = argc > 8 ? atoi(argv[8]) : 1,
            lone = argc > 9 ? atoi(argv[9]) : 0; /* 1: pipamd_engine_set_lone_batches (no second one-wave launch) */
  const size_t per = (size_t)B * ni * (nvar + 1);
  pipamd_batch_desc d = {B, nvar, 0, ni, -1, (nq ? PIPAMD_T_INT : 0) | PIPAMD_T_ROWS_STAY, nq ? ni + 64 : 0, 0, 64};
  int64_t *h = malloc(per * nb * sizeof(int64_t)), **rows = malloc(nb * sizeof *rows);
  uint64_t *piv_of = calloc(nb, sizeof *piv_of), *d_cnt, cnt[4];
  lane_t *L = calloc(K, sizeof *L);
  FILE *f = fopen(argv[1], "rb");
  if (!h || !rows || !piv_of || !L || !f || fread(h, sizeof(int64_t), per * nb, f) != per * nb) {
    fprintf(stderr, "cannot read %zu values from %s\n", per * nb, argv[1]);
    return 1;
  }
  fclose(f);
  HIP(hipSetDevice(0));
  for (int b = 0; b < nb; b++) {
    HIP(hipMalloc((void **)&rows[b], per * sizeof(int64_t)));
    HIP(hipMemcpy(rows[b], h + per * b, per * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  HIP(hipMalloc((void **)&d_cnt, 4 * sizeof(uint64_t)));
  for (int i = 0; i < K; i++) {
    CHECK(pipamd_engine_create(&L[i].e, 0));
    CHECK(pipamd_engine_set_timing(L[i].e, 0));
    CHECK(pipamd_engine_set_bulk_min(L[i].e, 256));
    if (lone) CHECK(pipamd_engine_set_lone_batches(L[i].e, 1));
    CHECK(pipamd_engine_set_max_rows(L[i].e, ni + 1024)); /* a tableau on which the cuts do not converge ends CAPACITY */
    HIP(hipStreamCreateWithFlags(&L[i].st, hipStreamNonBlocking));
    HIP(hipMalloc(&L[i].ws, pipamd_batch_workspace_bytes(&d)));
    HIP(hipMalloc((void **)&L[i].status, B * sizeof(int32_t)));
    HIP(hipMalloc((void **)&L[i].pivots, B * sizeof(int32_t)));
    HIP(hipMalloc((void **)&L[i].cuts, B * sizeof(int32_t)));
    HIP(hipMalloc((void **)&L[i].num, (size_t)B * nvar * sizeof(int64_t)));
    HIP(hipMalloc((void **)&L[i].den, (size_t)B * nvar * sizeof(int64_t)));
  }
  /* every batch once, synchronously: its pivot count (and a warm-up) */
  for (int b = 0; b < nb; b++) {
    lane_t *l = &L[b % K];
    CHECK(pipamd_batch_load(l->e, l->ws, &d, rows[b], l->st));
    CHECK(pipamd_batch_solve(l->e, l->ws, &d, l->st));
    CHECK(pipamd_batch_counters(l->e, l->ws, &d, d_cnt, l->st));
    HIP(hipStreamSynchronize(l->st));
    HIP(hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost));
    piv_of[b] = cnt[0];
  }
  /* the timed stream of batches */
  int next = 0, done = 0;
  unsigned long long total = 0;
  HIP(hipDeviceSynchronize());
  const double t0 = now_ms();
  while (done < steps) {
    for (int i = 0; i < K; i++) {
      lane_t *l = &L[i];
      if (l->busy) {
        const int r = pipamd_batch_poll(l->e); /* never blocks */
        if (r < 0) CHECK(r);
        if (r == 1) {
          CHECK(pipamd_batch_results(l->e, l->ws, &d, l->status, l->pivots, l->cuts, l->num, l->den, l->st));
          total += piv_of[l->batch];
          l->busy = 0;
          done++;
        }
      }
      if (!l->busy && next < steps) {
        l->batch = next++ % nb;
        CHECK(pipamd_batch_load(l->e, l->ws, &d, rows[l->batch], l->st));
        CHECK(pipamd_batch_solve_async(l->e, l->ws, &d, l->st));
        l->busy = 1;
      }
    }
  }
  HIP(hipDeviceSynchronize());
  const double ms = now_ms() - t0;
  /* how the last solve of every lane ended */
  long hist[10] = {0};
  int32_t *hs = malloc(B * sizeof(int32_t));
  for (int i = 0; i < K && i < steps; i++) {
    HIP(hipMemcpy(hs, L[i].status, B * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < B; k++) hist[hs[k] < 0 || hs[k] > 9 ? 8 : hs[k]]++;
  }
  printf("batch_stream: %d steps of %d tableaux on %d lanes, one host thread: %llu pivots in %.2f ms = %.1f M pivots/s;"
         " statuses of the lanes' last solves: solution %ld nil %ld other %ld\n",
         steps, B, K, total, ms, total / ms / 1e3, hist[PIPAMD_ST_SOLUTION], hist[PIPAMD_ST_NIL],
         hist[0] + hist[3] + hist[4] + hist[5] + hist[6] + hist[7] + hist[8] + hist[9]);
  for (int i = 0; i < K; i++) pipamd_engine_destroy(L[i].e);
  return 0;
}
