// piplib_amd/csrc/pip_host.h -- internal host-side declarations.
#ifndef PIP_HOST_H
#define PIP_HOST_H
#include <hip/hip_runtime.h>
#include <pthread.h>

#include "pip_job.h"

#define PIPAMD_MAX_ROUNDS 512
#define PIPAMD_MAX_GROW 24 /* growth rounds of one solve: the row capacity at least doubles per round, up to PIPAMD_SMAX */

struct pipamd_engine {
  int device;
  hipEvent_t ev[2 * PIPAMD_MAX_ROUNDS];
  int nev, nlaunch;
  int timed;
  int round_pivots;  /* pivot budget per tableau in the bulk launch (0 = default) */
  int round_rows;    /* spare rows (Gomory cuts) in the bulk launch's LDS image (0 = default) */
  int bulk_min;      /* batches of at least this many tableaux start with the one-wave bulk launch (0 = default 2048) */
  int single_launch; /* debug: stop after one launch */
  int lone_batches;  /* 1: no general one-wave launch between the lean launch and the tail (pipamd_engine_set_lone_batches) */
  int no_lean;       /* 1: bulk launches without the lean kernel (pip_lean.h) */
  int lean64;        /* 1: 128-bit batches of 129 ... 256 columns start with the lean kernel of pip_lean64.h (pipamd_engine_set_lean64) */
  int no_lean2;      /* 1: the second one-wave bulk launch is the general kernel even where the lean kernel could resume */
  int *h_run;        /* pinned: {jobs still running, their largest row count | PIPAMD_Q_CAPFLAG, how many of them are out of rows} */
  int *d_q;          /* launch-list control words (a pool, see pipamd_batch_solve) and the two job lists */
  unsigned solve_seq; /* solves since the control pool was last zeroed */
  hipStream_t pool_stream; /* the stream that zeroing was ordered on */
  int no_timing;     /* 1: no HIP events around the launches */
  int q_cap;         /* jobs the lists hold */
  int iter_limit;
  int waves_per_job; /* 0 = choose by batch size */
  int tail_waves;    /* waves per tableau of the tail launch when waves_per_job is 0 (0 = default 4) */
  int blocking_wait;   /* 1: pipamd_batch_solve naps between looks at its stream instead of spinning on it */
  int no_device_tree; /* 1: pipamd_solve_tableaux_lockstep skips the device-resident traiter() (pip_quast.hip) */
  void *dt_buf[8];     /* device tree: device buffers kept between calls (problems, rows, stacks, tapes, results, ...) */
  size_t dt_cap[8];
  int dt_small[8];     /* calls in a row that needed less than a quarter of the buffer (it is given back after eight) */
  pthread_mutex_t dt_lock; /* the device-tree buffers below serve one call at a time */
  void *dt_host;       /* device tree: pinned staging buffer for the problems' rows */
  size_t dt_host_cap;
  int dt_served, dt_fallback; /* problems the device tree finished / handed back in the last lock-step call */
  unsigned long long *d_prof; /* diagnostic builds only (-DPIP_PROFILE) */
  void *d_scratch;
  size_t scratch_bytes;
  /* expanser for the batch layer (pipamd_batch_solve): side arenas for tableaux that spent their spare rows, one
   * per growth round of a solve (a round doubles the row capacity), kept between solves; the block counter */
  void *d_side[PIPAMD_MAX_GROW];
  size_t side_bytes[PIPAMD_MAX_GROW];
  int *d_side_count;
  void *retired[128]; /* side arenas outgrown while batches were in flight: freed with the engine (hipFree waits for the device) */
  int nretired;
  int max_rows;      /* row budget per tableau of pipamd_batch_solve's growth (0 = the engine's limit) */
  int grow_step;     /* testing aid: rows added per growth round (0 = double) */
  int last_rehoused; /* tableaux the last pipamd_batch_solve re-housed (all rounds) */
  struct BatchRun *run; /* the batch solve in progress (pipamd_batch_solve_async .. pipamd_batch_wait) */
};

void pipamd_set_error(const char *fmt, ...);
int pipamd_batch_layout(const pipamd_batch_desc *d, PipBatchLayout *lay, size_t *jobs_bytes);

extern "C" {
/* bytes of the LDS image a launch over jobs of at most (Lmax, Smax, Wmax) needs; a job only fits
 * the engine if this stays within the 159 KiB a workgroup can get (PIPAMD_LDS_BUDGET) */
size_t pipk_advance_lds_bytes(int Lmax, int Smax, int Wmax, int ebits);
int pipk_static_class(int smax);
int pipk_lean_class(int smax);
hipError_t pipk_launch_advance(PipJob *jobs, long long *arena, int njobs, int Lmax, int Smax, int Wmax, int iter_limit,
                               int waves_per_job, int ebits, unsigned long long *prof, hipStream_t stream);
size_t pipk_lean64_lds_bytes(int Smax, int Lmax);
hipError_t pipk_launch_advance_q(PipJob *jobs, long long *arena, int njobs, int Lmax, int Smax, int Wmax, int iter_limit,
                                 int waves_per_job, int ebits, void *const *q5, int grid, void **big, int hints,
                                 unsigned long long *prof,
                                 hipStream_t stream);
hipError_t pipk_launch_replay_all(PipJob *jobs, long long *arena, int njobs, int ebits, int wave_per_job, hipStream_t stream);
hipError_t pipk_launch_batch_load(PipJob *jobs, long long *arena, const long long *rows, PipBatchLayout lay, int first,
                                  int count, hipStream_t stream);
hipError_t pipk_launch_batch_results(const PipJob *jobs, const long long *arena, int njobs, int nvar, int nparm,
                                     int ebits, int *status, int *pivots, int *cuts, void *sol_num, void *sol_den,
                                     hipStream_t stream);
hipError_t pipk_launch_rehouse(PipJob *jobs, long long *arena, void *const *q5, int grid, PipBatchLayout nl, int *side_count,
                               int side_cap, hipStream_t stream);
hipError_t pipk_launch_rehouse_finish(PipJob *jobs, long long *arena, int njobs, int sol_words, hipStream_t stream);
hipError_t pipk_launch_clone(long long *arena, const long long *list, int n, hipStream_t stream);
hipError_t pipk_launch_patch(long long *arena, const int *buf, const long long *index, int n, hipStream_t stream);
hipError_t pipk_launch_fresh(long long *arena, const long long *buf, const long long *index, int n, int ebits, hipStream_t stream);
hipError_t pipk_launch_gather(const PipJob *jobs, const long long *arena, int njobs, long long *out,
                              const long long *off, int ebits, hipStream_t stream);
hipError_t pipk_launch_batch_counters(const PipJob *jobs, int njobs, unsigned long long *out, hipStream_t stream);
}
#endif
