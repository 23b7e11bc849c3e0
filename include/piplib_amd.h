/* include/piplib_amd.h -- C ABI of the MI355X-native PipLib hot path.
 *
 * Plain C, plain pointers and sizes: this is what a PipLib maintainer binds to
 *  in place of the CPU implementations of
 *
 *   traiter()/pivoter()/choisir_piv()/exam_coef()   reference source/traiter.c:101-159,297-548,628-791
 *   integrer() (Gomory cut rows)                    reference source/integrer.c:305-534
 *   tab_alloc()/tab_get()/expanser() row store      reference source/tab.c:158-248, traiter.c:55-88
 *
 * Two public layers, lowest first:
 *   1. pipamd_batch_*   : a *uniform* batch of tableaux that lives in HBM; one
 *                         workgroup per tableau runs the whole pivot loop on the GPU.
 *   3. pipamd_traiter, pipamd_solve_*
 *                       : traiter() itself (traiter.c:628) -- the quast decision tree on the
 *                         host, every pivot on the GPU -- filling the reference's solution tape,
 *                         so that piplib.c / maind.c / sol.c stay the reference's own.
 * (Layer 2 -- heterogeneous jobs of any shape in one arena, advanced until each is finished or
 *  needs a host decision -- is internal: it is what layer 3's quast builder drives.)
 *
 * All device pointers are ordinary HIP device pointers (e.g. torch
 * `tensor.data_ptr()`); `stream` is a hipStream_t passed as void*.
 * Every function returns 0 on success or a negative PIPAMD_E_* code; nothing
 * here falls back to a CPU implementation.
 */
#ifndef PIPLIB_AMD_H
#define PIPLIB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface, also returned by pipamd_version() of the library that was loaded: a binding checks
 * the two against each other before it calls anything else (entry points were added and changed between versions
 * without the symbols changing).  100: round 1.  200: round 2 -- pipamd_traiter / pipamd_solve_tableau* hand out
 * tape cells (`cells, n_cells`) instead of text; the transliterated pip_solve front end (pipamd_pip_solve,
 * pipamd_quast_*) is gone: pip_solve stays the reference's piplib.c over bindings/piplib_traiter_hook.c.
 * 300: round 3 -- pipamd_batch_solve_async / _wait / _poll, pipamd_batch_load_part, unbounded row growth in the
 * batch layer (no PIPAMD_ST_CAPACITY short of the engine's 16,000-row limit), pipamd_solve_tableaux128,
 * pipamd_engine_set_max_rows.  400: round 4 -- pipamd_solve_tableaux_lockstep128, pipamd_engine_set_lean64; the 128-bit
 * entries try the device-resident traiter() first (pipamd_last_device_tree answers for them too); nothing removed. */
#define PIPAMD_VERSION 400

/* ---- error codes (return values) ---- */
#define PIPAMD_OK 0
#define PIPAMD_E_INVALID -1   /* bad argument / shape */
#define PIPAMD_E_HIP -2       /* a HIP runtime call failed (no GPU, OOM, ...) */
#define PIPAMD_E_TOOLARGE -3  /* shape exceeds the engine's compiled limits */
#define PIPAMD_E_NOMEM -4
#define PIPAMD_E_SOLVER -5    /* per-problem failure, see the status array */

/* ---- traiter flags (reference funcall.h:32-33) ---- */
#define PIPAMD_T_INT 1
#define PIPAMD_T_DUAL 2
/* engine-only flags */
#define PIPAMD_T_SORT 256    /* rows not yet sorted (tab_sort_rows, traiter.c:556) */
#define PIPAMD_T_DEEPEST 512 /* deepest-cut option (integrer.c:417-438) */
#define PIPAMD_T_NOSKIP 2048 /* measurement aid: rewrite every real row on every pivot, as the
                               reference's loop does (traiter.c:467-502); results are identical */
#define PIPAMD_T_ROWS_STAY 8192 /* pipamd_batch_desc.tflags: the caller keeps the `rows` array of pipamd_batch_load alive
                                  and unchanged until the batch's next pipamd_batch_solve has returned.  The load then
                                  only builds the row tables; the first pivot launch reads the rows from the caller's
                                  array in the pass that builds its summaries anyway (64-bit entries, an even number
                                  of columns, a 16-byte aligned array; otherwise the rows are copied as usual). */
#define PIPAMD_T_STATE 1024  /* a paused job's LDS summaries are saved in its state block */

/* ---- per-problem status written by the engine ---- */
#define PIPAMD_ST_RUN 0          /* not finished (iteration limit reached: relaunch) */
#define PIPAMD_ST_SOLUTION 1     /* all rows non-negative (and integral if T_INT): solution rows valid */
#define PIPAMD_ST_NIL 2          /* no solution (traiter.c:782-785, integrer.c:482-485) */
#define PIPAMD_ST_NEED_COMPA 3   /* parametric signs undecided: host runs compa_test (traiter.c:682) */
#define PIPAMD_ST_NEED_PARMCUT 4 /* parametric Gomory cut on row `aux`: host runs find/add_parm */
#define PIPAMD_ST_OVERFLOW 5     /* the reference's "Integer overflow" exit (traiter.c:424,442) */
#define PIPAMD_ST_CAPACITY 6     /* spare columns (cap_newparm) exhausted, or the engine's row limit (16,000 rows; 128-bit
                                    entries: what fits a workgroup's LDS) reached; pipamd_batch_solve itself grows rows */
#define PIPAMD_ST_RANGE 7        /* entries too large for the exact fast pivot-column choice */
#define PIPAMD_ST_INTERNAL 8
#define PIPAMD_ST_MAXCOL 9       /* "Too many variables" (integrer.c:324) */

/* ---- row flags (reference tab.h:55-62) ---- */
#define PIPAMD_F_UNIT 1
#define PIPAMD_F_PLUS 2
#define PIPAMD_F_MINUS 4
#define PIPAMD_F_ZERO 8
#define PIPAMD_F_CRITIC 16
#define PIPAMD_F_UNKNOWN 32

typedef struct pipamd_engine pipamd_engine;

/* Engine = device + limits.  `device` is a HIP ordinal (after HIP_VISIBLE_DEVICES). */
int pipamd_engine_create(pipamd_engine **out, int device);
void pipamd_engine_destroy(pipamd_engine *e);
const char *pipamd_last_error(void);
/* Upper bound on pivots per problem per launch (a problem still PIPAMD_ST_RUN afterwards is resumed by the next
 * launch of the same pipamd_batch_solve).  At most 512 (the default; larger values are clamped): a launch logs
 * (pivot, denominator) per pivot for the determinant bookkeeping of traiter.c:412-446, which a replay kernel runs
 * after the launch, and a job's log holds 512 entries.  pipamd_batch_solve gives up on a batch (PIPAMD_E_SOLVER)
 * after 512 launches, i.e. a tableau may take 262,144 pivots. */
int pipamd_engine_set_iter_limit(pipamd_engine *e, int pivots_per_launch);
/* Waves (64 lanes each) that share one tableau: 1 keeps more tableaux in flight per CU (best for
 * large batches of sparse problems), 4 spreads a tableau's rows over four waves (few or dense
 * tableaux), 8 likewise (64-bit entries of <= 128 columns); 0 (default) = 1 when the batch has >= 2048
 * tableaux, else 4. */
int pipamd_engine_set_waves_per_job(pipamd_engine *e, int waves);
/* Waves per tableau of the tail launch of pipamd_batch_solve (4, the default, or 8).  Eight spread
 * the 15-25 rows a late pivot of a long tableau rewrites over twice the waves: a caller that runs
 * one batch at a time gains ~8 %; with many batches in flight the extra waves of a tail crowd out
 * other batches' bulk launches (-3 %).  64-bit entries of <= 128 columns; 4 otherwise. */
int pipamd_engine_set_tail_waves(pipamd_engine *e, int waves);
/* One batch at a time (1) or many in flight on other engines (0, the default).  A bulk-sized batch of 127 unknowns in
 * 64 bits starts with the lean one-wave launch (entries below 2^15); the tableaux that launch leaves go through the
 * general one-wave launch and then the tail launches.  That middle launch keeps the device full when other batches'
 * launches run beside it; for a lone batch it is a long, nearly empty launch, and the tableaux go straight to the
 * tail launches instead (a lone 10k batch: 4.7 ms instead of 6.3 ms; 14 batches in flight: 10 % fewer pivots/s). */
int pipamd_engine_set_lone_batches(pipamd_engine *e, int on);
/* 128-bit batches without parameters of 129 ... 256 columns (at least 128 tableaux): 1 = the first launch of
 * pipamd_batch_solve is the lean kernel of csrc/pip_lean64.h -- one wave per tableau, rows held as long longs while every
 * entry fits 63 bits (half the traffic and registers), tableaux with a wider entry handed over to the 128-bit kernel.
 * Default 0: on BASELINE's configs[4] it is no faster than the four-waves-per-tableau 128-bit kernel. */
int pipamd_engine_set_lean64(pipamd_engine *e, int on);
/* A tail launch of a 128-bit batch over at most 256 tableaux gives each a whole CU (sixteen waves): the hundreds of rows a
 * late pivot of a long tableau rewrites spread over four times the waves (built in; pipamd_engine_set_tail_waves or
 * _set_waves_per_job switch it off). */
/* How pipamd_batch_solve waits for the device at its end: 0 (default) polls the stream, which is the
 * quickest for a few host threads; 1 naps 40 us between looks at the stream.  With more batches in flight
 * than the host has CPUs the spinning threads take turns on the cores: 48 batches of 1,250 tableaux on 16 CPUs ran
 * at 157 M pivots/s polling and 240 M sleeping. */
int pipamd_engine_set_blocking_wait(pipamd_engine *e, int on);
int pipamd_version(void);

/* ------------------------------------------------------------------ layer 1 */
typedef struct pipamd_batch_desc {
  int32_t batch;       /* number of tableaux */
  int32_t nvar;        /* unknowns */
  int32_t nparm;       /* parameters (0 => whole solve stays on the GPU) */
  int32_t ni;          /* inequality rows per tableau */
  int32_t bigparm;     /* column index of the big parameter or -1 */
  int32_t tflags;      /* PIPAMD_T_INT ... */
  int32_t cap_cuts;    /* spare rows per tableau for Gomory cuts */
  int32_t cap_newparm; /* spare columns per tableau (parametric cuts) */
  int32_t entier_bits; /* 0 or 64: int64 entries (the reference's long long build);
                          128: __int128 entries, same algorithm, overflow-safe variant */
} pipamd_batch_desc;

/* bytes of device workspace the batch needs (tableaux + row tables + job table + results) */
size_t pipamd_batch_workspace_bytes(const pipamd_batch_desc *d);

/* tab_get() for a whole batch: rows[b][i][0..ncol) (ncol = nvar+nparm+1, PIP column
 * order unknowns|constant|parameters) become Unknown rows with denominator 1 under nvar
 * unit rows.  `d_rows` is device memory, int64. */
int pipamd_batch_load(pipamd_engine *e, void *d_workspace, const pipamd_batch_desc *d,
                      const int64_t *d_rows, void *stream);

/* The same for the tableaux first .. first + count - 1 of the batch only; `d_rows` holds those `count` tableaux.  A
 * batch can thus be assembled from several row arrays -- e.g. one GPU's shards of the K batches a caller has in
 * flight, fused into one workspace so that ONE launch sequence serves them all instead of K small ones (what
 * bench.py's strong-scaling mode does; piplib_amd/dist.py).  Every tableau of the batch must have been loaded by
 * some call before pipamd_batch_solve. */
int pipamd_batch_load_part(pipamd_engine *e, void *d_workspace, const pipamd_batch_desc *d, const int64_t *d_rows,
                           int first, int count, void *stream);

/* traiter() on every tableau of the batch.  Launches on `stream` and returns when every
 * tableau has a final status (it synchronises the stream between rounds).  A tableau that spends its cap_cuts
 * spare rows is moved to a block of twice the row capacity (the reference's expanser, traiter.c:55-88, as integrer
 * calls it on a full tableau, integrer.c:410-415) as often as it takes, up to the engine's 16,000 rows: cap_cuts
 * is a size hint, not a limit. */
int pipamd_batch_solve(pipamd_engine *e, void *d_workspace, const pipamd_batch_desc *d, void *stream);

/* The same in two halves, so that ONE host thread keeps many batches in flight: pipamd_batch_solve_async enqueues
 * the launch sequence (bulk launch, tail launch, a copy of the tail's control words) on `stream` and returns without
 * waiting.  pipamd_batch_poll never blocks: 0 while the launches enqueued so far run; once they have ended it looks at
 * what they left -- if tableaux remain (beyond the per-launch pivot limit, or out of spare rows: those are re-housed)
 * it enqueues the next launches and returns 0 again; 1 when every tableau has its final status (or nothing is in
 * flight); < 0 on error.  pipamd_batch_wait does the same blocking and returns PIPAMD_OK when the batch is done.
 * One solve in flight per engine -- an engine is a small host object: a caller keeps K of them, each with its own
 * stream and workspace, starts a batch on each and polls them in turn, starting the next batch on whichever is done:
 * a lone batch leaves most of the GPU idle in its latency-bound tail, K batches in different phases fill it
 * (bench.py: one host thread, K = 14).  Results may be fetched once poll has returned 1 / wait has returned. */
int pipamd_batch_solve_async(pipamd_engine *e, void *d_workspace, const pipamd_batch_desc *d, void *stream);
/* Row budget of the growth above: a tableau is re-housed only while its row capacity stays within `rows` (at least
 * ni + cap_cuts; 0 = the default, only the engine's own limit) and ends PIPAMD_ST_CAPACITY beyond it.  The reference
 * grows without bound and so does the default; on inputs where Gomory cuts converge slowly (thousands of cut rows,
 * every pivot then rewriting thousands of rows) a caller bounds the memory and time of a batch with it. */
int pipamd_engine_set_max_rows(pipamd_engine *e, int rows);
int pipamd_batch_wait(pipamd_engine *e);
int pipamd_batch_poll(pipamd_engine *e);

/* Copy out, device to device: status[b], pivots[b], cuts[b] (int32 each, may be NULL),
 * sol_num[b][i][0..nparm] (parameter coefficients then constant, as solution() emits them,
 * traiter.c:255-271) and sol_den[b][i], i < nvar: int64 each, or little-endian pairs of
 * int64 (low, high) per value when entier_bits == 128. */
int pipamd_batch_results(pipamd_engine *e, const void *d_workspace, const pipamd_batch_desc *d,
                         int32_t *d_status, int32_t *d_pivots, int32_t *d_cuts, int64_t *d_sol_num,
                         int64_t *d_sol_den, void *stream);

/* Batch totals, device memory, 4 x uint64: [0] pivots (calls of pivoter), [1] Gomory cuts,
 * [2] rows rewritten by pivots (rows whose pivot-column entry is zero and that are already
 * reduced are left untouched -- same result as the reference's multiply-by-one pass),
 * [3] tableaux finished (solution or nil). */
int pipamd_batch_counters(pipamd_engine *e, const void *d_workspace, const pipamd_batch_desc *d,
                          uint64_t *d_out4, void *stream);

/* Bytes ONE pivot of one tableau of this shape moves when every real row is read and written,
 * as the reference's loop does (traiter.c:467-502): 2 * ni * ncol * sizeof(Entier).  The engine
 * itself only touches the rows that change; bench.py's `roofline` uses that smaller figure --
 * 8 * ncol * (2 * rows_rewritten + 2 * pivots) from pipamd_batch_counters -- and reports this one
 * as `dense_equivalent_GBps` and for `roofline_dense_mode` (PIPAMD_T_NOSKIP). */
size_t pipamd_dense_pivot_bytes(const pipamd_batch_desc *d);

/* Sum of the pivot-kernel launch durations of the last pipamd_batch_solve in milliseconds,
 * measured with HIP events on the launch stream, and the number of those launches. */
int pipamd_last_solve_ms(pipamd_engine *e, float *ms);
/* The events cost four runtime calls per solve, which shows on batches of a few hundred small
 * tableaux: off = pipamd_batch_solve records none (pipamd_last_solve_ms then fails).  Default on. */
int pipamd_engine_set_timing(pipamd_engine *e, int on);
int pipamd_last_solve_launches(pipamd_engine *e);
/* pipamd_batch_solve serves a batch of >= 2048 tableaux with two queue-fed launches and no host
 * round trip in between: a bulk launch of persistent one-wave workgroups that draw tableaux from
 * a device queue and give one up when it is finished, has used `pivots` pivots (default 96) or
 * `rows` Gomory-cut rows (default 48: the bulk launch's LDS image holds the input rows plus that
 * many, so that 24 tableaux fit a CU), and a tail launch that runs what is left to completion
 * with four waves per tableau. */
int pipamd_engine_set_round_pivots(pipamd_engine *e, int pivots);
int pipamd_engine_set_round_rows(pipamd_engine *e, int rows);
/* Smaller batches skip the bulk launch (few tableaux fill the GPU better with four waves each).
 * A caller that keeps several small batches in flight on separate streams -- together they do fill
 * the GPU -- lowers the threshold (default 2048 tableaux). */
int pipamd_engine_set_bulk_min(pipamd_engine *e, int tableaux);

/* ------------------------------------------------------------------ layer 3 */
/* The solution tape.  traiter() does not return a value: it pushes cells onto the tape of
 * source/sol.c (struct S, sol.c:52-59: flags, param1, param2) through sol_nil / sol_if / sol_list /
 * sol_forme / sol_new / sol_div / sol_val (sol.c:104-209), and the reference's front ends read
 * that tape afterwards: sol_edit (sol.c:291) prints it, sol_quast_edit (sol.c:664) turns it into a
 * PipQuast.  The engine hands the same cells out, in the order the reference would have pushed
 * them, so those readers keep working unchanged (INTEGRATION.md shows the few lines that replay
 * the cells into sol.c; bindings/piplib_traiter_hook.c is that code, compiled and run by the tests). */
#define PIPAMD_SOL_NIL 1  /* sol_nil()                       sol.c:42-50 */
#define PIPAMD_SOL_IF 2   /* sol_if()                                    */
#define PIPAMD_SOL_LIST 3 /* sol_list(param1)                            */
#define PIPAMD_SOL_FORM 4 /* sol_forme(param1)                           */
#define PIPAMD_SOL_NEW 5  /* sol_new(param1)                             */
#define PIPAMD_SOL_DIV 6  /* sol_div()                                   */
#define PIPAMD_SOL_VAL 7  /* sol_val(param1, param2): numerator, denominator, not reduced */
typedef struct pipamd_sol_cell {
  int32_t kind, reserved;
  int64_t param1, param2;
} pipamd_sol_cell;

/* traiter(tp, ctxt, nvar, nparm, ni, nc, bigparm, flags) -- reference source/traiter.c:628, as
 * called from pip_solve (piplib.c:858), maind.c:205 and for the empty-context test
 * (piplib.c:848, maind.c:198).  `tableau` = the ni inequality rows of `tp` (host, row-major,
 * nvar+nparm+1 int64 each, PIP column order unknowns|constant|parameters; Unknown rows with
 * denominator 1, as tab_get / tab_Matrix2Tableau build them), `context` = the nc rows of `ctxt`
 * (nparm+1 each), flags = 0, PIPAMD_T_INT or PIPAMD_T_DUAL (funcall.h:32-33), deepest_cut = the
 * reference's global of that name (piplib.c:53).  The quast decision tree runs on the host, every
 * pivot on the GPU.  On success *cells is a malloc'ed array of *n_cells cells (free with
 * pipamd_free).  Where the reference would have exit()ed ("Integer overflow", ...) the call
 * returns PIPAMD_E_SOLVER and *status holds the PIPAMD_ST_* reason. */
int pipamd_traiter(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int flags, int deepest_cut,
                   const int64_t *tableau, const int64_t *context, pipamd_sol_cell **cells, size_t *n_cells,
                   int *status, int64_t *pivots);

/* The same on 128-bit entries -- the reference's overflow-safe flavour is a whole-library build
 * with a wider Entier (include/piplib/piplib.h:42-88): device tableaux, context, parametric cuts
 * and tape all carry __int128 here; the input rows are int64, the cells' parameters come back as
 * (low, high) int64 pairs.  Tableaux must fit a workgroup's LDS (about 1,600 rows of <= 128
 * columns).  Small problems run their whole decision tree on the device in this flavour too (the 128-bit instantiation
 * of csrc/pip_quast.hip), many problems go through pipamd_solve_tableaux128 / pipamd_solve_tableaux_lockstep128. */
typedef struct pipamd_sol_cell128 {
  int32_t kind, reserved;
  int64_t param1_lo, param1_hi, param2_lo, param2_hi;
} pipamd_sol_cell128;
int pipamd_traiter128(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int flags, int deepest_cut,
                      const int64_t *tableau, const int64_t *context, pipamd_sol_cell128 **cells, size_t *n_cells,
                      int *status, int64_t *pivots);
int pipamd_solve_tableau128(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int nq,
                            const int64_t *ineq, const int64_t *ctx, int simplify, int deepest_cut,
                            pipamd_sol_cell128 **cells, size_t *n_cells, int *status, int64_t *pivots);

/* One problem in PIP's native tableau form (what maind.c reads from a .dat file): the whole of
 * maind.c:190-231 -- tab_simplify (tab.c:396) first when nq != 0 and `simplify`, the
 * empty-context test, then traiter.  An empty tape (*n_cells == 0) with PIPAMD_OK is maind.c's
 * "void" (empty context). */
int pipamd_solve_tableau(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int nq,
                         const int64_t *ineq, const int64_t *ctx, int simplify, int deepest_cut,
                         pipamd_sol_cell **cells, size_t *n_cells, int *status, int64_t *pivots);
void pipamd_free(void *p);

/* The same for `n` independent problems: `nthreads` host threads, each with its own decision
 * tree, device arena and HIP stream, share the GPU (their launches overlap).  cells[i] /
 * n_cells[i] / rcs[i] / statuses[i] / pivots[i] are what pipamd_solve_tableau returns for
 * problem i (statuses and pivots may be NULL). */
typedef struct pipamd_problem {
  int32_t nvar, nparm, ni, nc, bigparm, nq;
  const int64_t *ineq; /* ni x (nvar+nparm+1) */
  const int64_t *ctx;  /* nc x (nparm+1) */
} pipamd_problem;
int pipamd_solve_tableaux(pipamd_engine *e, int n, const pipamd_problem *problems, int simplify,
                          int deepest_cut, int nthreads, pipamd_sol_cell **cells, size_t *n_cells, int *rcs,
                          int *statuses, int64_t *pivots);

/* The same on 128-bit entries (small problems on the device first, then one TreeT<__int128> per host thread; cells as
 * pipamd_traiter128 hands them out); the lock-step scheduler has a 128-bit entry too (pipamd_solve_tableaux_lockstep128). */
int pipamd_solve_tableaux128(pipamd_engine *e, int n, const pipamd_problem *problems, int simplify,
                             int deepest_cut, int nthreads, pipamd_sol_cell128 **cells, size_t *n_cells, int *rcs,
                             int *statuses, int64_t *pivots);

/* The same results from a lock-step scheduler: one explicit traiter() state machine per problem
 * and, per step, ONE clone / patch / pivot-kernel / gather sequence for the whole batch, so the
 * host <-> device latency is paid once per step instead of once per problem.  Problems that need
 * a rare path (tableau growth beyond the reserved block, deepest cuts) are finished by the
 * per-problem tree of pipamd_solve_tableau. */
int pipamd_solve_tableaux_lockstep(pipamd_engine *e, int n, const pipamd_problem *problems, int simplify,
                                   int deepest_cut, pipamd_sol_cell **cells, size_t *n_cells, int *rcs,
                                   int *statuses, int64_t *pivots);
/* The lock-step scheduler of the overflow-safe flavour (piplib.h:42-88, funcall.h:37-41: the flavour is a type choice):
 * device tableaux, contexts, parametric cuts and tape cells are 128-bit, one launch sequence per step serves the whole
 * batch; no device-resident traiter() in front of it (that kernel is 64-bit); rare paths go to a TreeT<__int128>. */
int pipamd_solve_tableaux_lockstep128(pipamd_engine *e, int n, const pipamd_problem *problems, int simplify,
                                      int deepest_cut, pipamd_sol_cell128 **cells, size_t *n_cells, int *rcs,
                                      int *statuses, int64_t *pivots);
/* Small problems (at most 64 columns, spare room for new parameters included, and 104 inequalities -- 128 real rows with
 * the cuts; round 4: in both entry widths) are first given to the device-resident traiter() (csrc/pip_quast.hip): one wave per
 * problem runs the whole call tree -- pivots, compa_test sub-problems (traiter.c:162-243), forks of the
 * quast (traiter.c:695-759), cuts with new parameters (integrer.c:156-291) -- and writes the tape, with
 * no host round trip.  A problem in which a 64-bit operation would overflow, or that outgrows its
 * reserved rows, is handed back and served by the lock-step scheduler / the per-problem tree, which
 * reproduce the reference's wrap-around and "Integer overflow" behaviour.  pipamd_traiter,
 * pipamd_solve_tableau and pipamd_solve_tableaux try it first too (since interface version 300 also with
 * PIPAMD_T_DUAL), and so do their 128-bit counterparts with the kernel's 128-bit instantiation (there every product and
 * sum is checked against 128 bits).  On by default (the environment
 * variable PIPAMD_NO_DEVICE_TREE switches it off for a process);
 * pipamd_last_device_tree reports how many problems of the last lock-step call each side served. */
int pipamd_engine_set_device_tree(pipamd_engine *e, int on);
int pipamd_last_device_tree(const pipamd_engine *e, int *served, int *handed_back);

#ifdef __cplusplus
}
#endif
#endif /* PIPLIB_AMD_H */
