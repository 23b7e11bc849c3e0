// piplib_amd/csrc/pip_job.h -- structures shared by the HIP kernels and the host side.
#ifndef PIP_JOB_H
#define PIP_JOB_H
#include <stdint.h>

#include "../../include/piplib_amd.h"

#define PIPAMD_MAXDET 4   /* reference tab.h:67 MAX_DETERMINANT */
#define PIPAMD_MAXCOL 512 /* reference type.h:44 */
#define PIPAMD_MAXPARM 50 /* reference type.h:45 */
/* Row tables are indexed with 16-bit codes (slot < 0x4000, logical row < 0xffff).  A job whose row
 * tables fit a workgroup's LDS (about 3,400 rows of <= 128 int64 columns) is staged there; a larger
 * 64-bit job runs with the same tables in HBM (the kernel's GM instantiation), so that a tableau
 * can grow as with the reference's expanser (traiter.c:55-88) up to these limits. */
#define PIPAMD_SMAX 16000 /* real rows (slots) per job */
#define PIPAMD_LMAX (PIPAMD_SMAX + PIPAMD_MAXCOL) /* logical rows */
#define PIPAMD_LDS_BUDGET (160 * 1024 - 1024) /* dynamic LDS a workgroup can get */
/* The pivot kernel does not run the determinant bookkeeping of traiter.c:412-446 itself
 * (wave-uniform scalar work, ~12 % of its instructions with 64-bit entries); it logs (pivot,
 * denominator of the pivot row) per pivot -- at most this many per launch -- and pip_det_replay_kernel replays
 * the log right after the launch, one wave per job: the gcds of 64 pivots at a time on the lanes,
 * only the walk over the limbs sequentially. */
#define PIPAMD_DETLOG 512

/* One problem ("job") in the device arena.  All offsets are in int64 units from the
 * arena base and are even (rows are 16-byte aligned).
 *   rows_off: den[L] (int64) | flag[L] (int32) | ref[L] (int32)      = 2*L int64
 *   vals_off: S slots of W int64 (zero beyond the live columns)
 *   sol_off : nvar*(nparm+1) numerators | nvar denominators
 *   state_off: LDS summaries of a paused job: nzm[S][NM] (u64) | sig[L] (u16) | rbits[L] (u8)
 * Mirrors the reference's struct T / struct L (tab.h:36-85) without pointers. */
/* internal tflags bit: the job's rows still sit in the caller's array (src_rows); the first pivot launch
 * reads them from there while it builds its summaries and writes them into the block (PIPAMD_T_ROWS_STAY) */
#define PIPAMD_T_FRESHROWS 4096

typedef struct PipJob {
  int64_t vals_off, rows_off, sol_off, state_off;
  int64_t log_off; /* 2 * PIPAMD_DETLOG entries: the determinant log of the last launch */
  int32_t nvar, nparm, ni, bigparm;
  int32_t tflags;
  int32_t L, S, W;
  int32_t status, aux, npiv, ncut;
  int32_t ldet, nupd; /* nupd: rows rewritten by pivots so far (excludes skipped zero-multiplier rows) */
  int64_t det[2 * PIPAMD_MAXDET]; /* multi-limb determinant, tab.h:76-81: det[i] (64-bit entries) or
                                     det[2i] | det[2i+1] << 64 (128-bit entries) */
  uint64_t maxabs;
  int32_t nlog, pad_; /* entries of the determinant log not replayed yet */
  int32_t state_nch, ebits; /* ebits: 64 or 128 (0 = 64) */ /* row-chunk count (NCH) of the launch that saved the state block */
  int64_t src_rows; /* PIPAMD_T_FRESHROWS: device address of the caller's ni x ncol input rows (not yet in the block) */
  int64_t home_sol_off; /* != 0: the job was re-housed in a larger block outside its batch's workspace (expanser,
                           pip_rehouse_kernel); its solution is copied back to this offset when the solve ends */
} PipJob;

/* pipamd_batch_solve's launch lists: a job that ran out of spare rows (PIPAMD_ST_CAPACITY) stays on the list, and
 * the list's `maxni` word carries this bit, until the host has re-housed it in a larger block; the word behind `maxni`
 * counts these jobs (the host sizes the larger blocks' arena by it) */
#define PIPAMD_Q_CAPFLAG (1 << 30)

typedef struct PipBatchLayout {
  int64_t arena_off; /* first job's block, int64 units */
  int64_t per_job;   /* block size per job, int64 units */
  int32_t batch, nvar, nparm, ni, bigparm, tflags;
  int32_t L, S, W;
  int32_t sol_words, state_words;
  int32_t ebits, pad;
} PipBatchLayout;

#endif
