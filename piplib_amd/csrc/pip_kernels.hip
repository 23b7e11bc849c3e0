// piplib_amd/csrc/pip_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X).
//
// The pivot kernel itself (pip_advance_kernel: traiter() / pivoter() / choisir_piv() / exam_coef() / integrer() /
// tab_sort_rows of the reference) lives in pip_advance.h and is instantiated by the pip_adv_*.hip files; this file
// holds the determinant replay, the batch load / results / counters kernels, expanser for the batch layer, the
// helpers of the lock-step scheduler and every launcher.
#include "pip_lean.h"
#include "pip_lean64.h"

// the instantiations of the pivot kernel's launcher live in pip_adv_*.hip
#define PIP_ADV_EXTERN(...) extern template hipError_t launch_advance_t<__VA_ARGS__>(const AdvanceLaunch &);
PIP_ADV_GROUP_A(PIP_ADV_EXTERN)
PIP_ADV_GROUP_B(PIP_ADV_EXTERN)
PIP_ADV_GROUP_C(PIP_ADV_EXTERN)
PIP_ADV_GROUP_D(PIP_ADV_EXTERN)
PIP_ADV_GROUP_F(PIP_ADV_EXTERN)
#undef PIP_ADV_EXTERN
#define PIP_LEAN_EXTERN(SC, FULL) extern template hipError_t launch_lean<SC, FULL>(const AdvanceLaunch &);
PIP_LEAN_CLASSES(PIP_LEAN_EXTERN)
#undef PIP_LEAN_EXTERN

// ------------------------------------------------------------- determinant replay
// traiter.c:394-446 for the pivots a launch logged: d = gcd(pivot, dpiv); the limbs lose the
// factors of dpiv / d (if some factor is left: "Integer overflow"); the first limb with room takes
// pivot / d (a fourth limb: "Integer overflow").  The bookkeeping never feeds back into the pivot
// loop, so it can trail it; an overflow found here overrides whatever the job's status became
// and sets the pivot count to the pivot that overflowed.  In the pivot loop itself this was
// wave-uniform scalar work: ~12 % of all instructions with 64-bit entries, ~18 % of the run time
// with 128-bit ones.
template <class T>
__device__ __forceinline__ T det_limb(const PipJob *J, int i) {
  if constexpr (sizeof(T) == 16)
    return (T)(((u128)(u64)J->det[2 * i + 1] << 64) | (u64)J->det[2 * i]);
  else
    return (T)J->det[i];
}
template <class T>
__device__ __forceinline__ void det_limb_store(PipJob *J, int i, T x) {
  if constexpr (sizeof(T) == 16) {
    J->det[2 * i] = (i64)(u64)(u128)x;
    J->det[2 * i + 1] = (i64)(u64)((u128)x >> 64);
  } else
    J->det[i] = (i64)x;
}
// one step of the walk over the limbs for the reduced pair (ppivot, dppiv); false = overflow
template <class T>
__device__ __forceinline__ bool det_step(T &det0, T &det1, T &det2, T &det3, int &ldet, T ppivot, T dppiv) {
  // once dppiv is 1 the remaining limbs would be divided by gcd(limb, 1) = 1
#define PIP_DET_DIVIDE(limb, i)                    \
  if ((i) < ldet && dppiv != 1) {                  \
    const T d_ = gcd_i64(limb, dppiv);             \
    if (d_ != 1) {                                 \
      limb = exact_quo(limb, d_);                  \
      dppiv = exact_quo(dppiv, d_);                \
    }                                              \
  }
  PIP_DET_DIVIDE(det0, 0)
  PIP_DET_DIVIDE(det1, 1)
  PIP_DET_DIVIDE(det2, 2)
  PIP_DET_DIVIDE(det3, 3)
#undef PIP_DET_DIVIDE
  if (dppiv != 1) return false;
  constexpr int B = ET<T>::BITS;
  const int lp = log2_64(ppivot);
  if (0 < ldet && log2_64(det0) + lp < B)
    det0 = wmul(det0, ppivot);
  else if (1 < ldet && log2_64(det1) + lp < B)
    det1 = wmul(det1, ppivot);
  else if (2 < ldet && log2_64(det2) + lp < B)
    det2 = wmul(det2, ppivot);
  else if (3 < ldet && log2_64(det3) + lp < B)
    det3 = wmul(det3, ppivot);
  else {
    ldet++;
    if (ldet >= PIPAMD_MAXDET) return false;
    if (ldet == 1)
      det0 = ppivot;
    else if (ldet == 2)
      det1 = ppivot;
    else
      det2 = ppivot;
  }
  return true;
}

// One WAVE per job: gcd(pivot, dpiv) and the two quotients do not depend on the limbs, so the
// lanes compute them for 64 pivots at once; only the walk over the limbs is sequential, on the
// scalar unit.  Shortest latency: used where few jobs ran and someone waits for them.
template <class T>
__global__ __launch_bounds__(64) void pip_det_replay_kernel(PipJob *jobs, i64 *arena, int njobs, PipQueue q) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const int nq = q.in_count ? *q.in_count : njobs;
  if (t >= nq) return;
  PipJob *J = &jobs[q.in_list ? q.in_list[t] : t];
  const int nlog = J->nlog;
  if (nlog <= 0 || (J->ebits == 128) != (sizeof(T) == 16)) return;
  const T *lg = (const T *)(arena + J->log_off);
  T det0 = uni64(det_limb<T>(J, 0)), det1 = uni64(det_limb<T>(J, 1)), det2 = uni64(det_limb<T>(J, 2)),
    det3 = uni64(det_limb<T>(J, 3));
  int ldet = __builtin_amdgcn_readfirstlane(J->ldet);
  int bad = -1;
  for (int base = 0; base < nlog && bad < 0; base += 64) {
    const int k = base + lane;
    T pp = 1, dp = 1;
    if (k < nlog) {
      pp = lg[2 * k];
      dp = lg[2 * k + 1];
      if (dp != 1) {
        const T d = gcd_i64(pp, dp);
        if (d != 1) {
          pp = exact_quo(pp, d);
          dp = exact_quo(dp, d);
        }
      }
    }
    const int n = nlog - base < 64 ? nlog - base : 64;
    for (int j = 0; j < n; j++) {
      if (!det_step<T>(det0, det1, det2, det3, ldet, readlane64(pp, j), readlane64(dp, j))) {
        bad = base + j;  // traiter.c:424,442: the reference exits inside this call of pivoter
        break;
      }
    }
  }
  if (lane == 0) {
    if (bad >= 0) {
      J->status = PIPAMD_ST_OVERFLOW;
      J->npiv = J->npiv - nlog + bad + 1;
    }
    det_limb_store<T>(J, 0, det0);
    det_limb_store<T>(J, 1, det1);
    det_limb_store<T>(J, 2, det2);
    det_limb_store<T>(J, 3, det3);
    J->ldet = ldet;
    J->nlog = 0;
  }
}

// One LANE per job (64 jobs per wave): far fewer instructions issued in all -- what counts behind
// a bulk launch, when the GPU is kept busy by other batches -- at the price of a longer latency
// per job (a lane walks its log alone).
template <class T>
__global__ __launch_bounds__(64) void pip_det_replay_lanes_kernel(PipJob *jobs, i64 *arena, int njobs, PipQueue q) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = q.in_count ? *q.in_count : njobs;
  if (t >= nq) return;
  PipJob *J = &jobs[q.in_list ? q.in_list[t] : t];
  const int nlog = J->nlog;
  if (nlog <= 0 || (J->ebits == 128) != (sizeof(T) == 16)) return;
  const T *lg = (const T *)(arena + J->log_off);
  T det0 = det_limb<T>(J, 0), det1 = det_limb<T>(J, 1), det2 = det_limb<T>(J, 2), det3 = det_limb<T>(J, 3);
  int ldet = J->ldet;
  constexpr int CH = sizeof(T) == 16 ? 4 : 8;  // log entries per 128-byte line
  bool stop = false;
  for (int k0 = 0; k0 < nlog && !stop; k0 += CH) {
    T ev[CH][2];
#pragma unroll
    for (int u = 0; u < CH; u++) {
      const bool in = k0 + u < nlog;
      ev[u][0] = in ? lg[2 * (k0 + u)] : (T)1;
      ev[u][1] = in ? lg[2 * (k0 + u) + 1] : (T)1;
    }
#pragma unroll
    for (int u = 0; u < CH; u++) {
      const int k = k0 + u;
      if (k >= nlog || stop) break;
      T ppivot = ev[u][0], dppiv = ev[u][1];
      if (dppiv != 1) {
        const T d = gcd_i64(ppivot, dppiv);
        if (d != 1) {
          ppivot = exact_quo(ppivot, d);
          dppiv = exact_quo(dppiv, d);
        }
      }
      if (!det_step<T>(det0, det1, det2, det3, ldet, ppivot, dppiv)) {
        J->status = PIPAMD_ST_OVERFLOW;
        J->npiv = J->npiv - nlog + k + 1;
        stop = true;
      }
    }
  }
  det_limb_store<T>(J, 0, det0);
  det_limb_store<T>(J, 1, det1);
  det_limb_store<T>(J, 2, det2);
  det_limb_store<T>(J, 3, det3);
  J->ldet = ldet;
  J->nlog = 0;
}

// ---------------------------------------------------------------- batch load
// tab_alloc + tab_get (tab.c:158-248) for a uniform batch: nvar unit rows, then
// ni Unknown rows with denominator 1; spare slots and columns zeroed.  Input rows are int64
// whatever the entry type of the tableau.
template <class T>
__global__ void pip_batch_load_kernel(PipJob *jobs, i64 *arena, const i64 *rows, PipBatchLayout lay, int first) {
  constexpr int EW = ET<T>::EW;
  const int b = first + blockIdx.x;  // `rows` holds the tableaux first, first + 1, ... of the batch
  const int tid = threadIdx.x;
  const int ncol = lay.nvar + lay.nparm + 1;
  PipJob *J = &jobs[b];
  const int64_t base = lay.arena_off + (int64_t)b * lay.per_job;
  T *g_den = (T *)(arena + base);
  int *g_flag = (int *)(g_den + lay.L);
  int *g_ref = g_flag + lay.L;
  const int64_t rows_words = (int64_t)lay.L * EW + lay.L;
  T *vals = (T *)(arena + base + rows_words);
  for (int i = tid; i < lay.L; i += blockDim.x) {
    if (i < lay.nvar) {
      g_flag[i] = PIPAMD_F_UNIT;
      g_ref[i] = i;
      g_den[i] = 1;
    } else if (i < lay.nvar + lay.ni) {
      g_flag[i] = PIPAMD_F_UNKNOWN;
      g_ref[i] = i - lay.nvar;
      g_den[i] = 1;
    } else {
      g_flag[i] = 0;
      g_ref[i] = 0;
      g_den[i] = 0;
    }
  }
  // input rows; spare slots only need their columns beyond ncol cleared (a cut row writes its
  // first ncol columns itself, a parametric cut relies on the new column being 0 elsewhere)
  const i64 *src = rows + (size_t)blockIdx.x * lay.ni * ncol;
  const int pad = lay.W - ncol;
  const bool defer = lay.pad != 0 && EW == 1;  // PIPAMD_T_ROWS_STAY: the first pivot launch fetches the rows itself
  if (!defer) {
    // (row, column) advance with the thread stride: no division per element
    const int ds = (int)blockDim.x / lay.W, dj = (int)blockDim.x % lay.W;
    int s = tid / lay.W, j = tid % lay.W;
    for (int e = tid; e < lay.ni * lay.W; e += blockDim.x) {
      vals[e] = j < ncol ? (T)src[(size_t)s * ncol + j] : (T)0;
      s += ds;
      j += dj;
      if (j >= lay.W) {
        j -= lay.W;
        s++;
      }
    }
  }
  const int first_pad_row = defer ? 0 : lay.ni;
  for (int e = tid; e < (lay.S - first_pad_row) * pad; e += blockDim.x) {
    int s = first_pad_row + e / pad, j = ncol + e % pad;
    vals[(size_t)s * lay.W + j] = 0;
  }
  if (tid == 0) {
    J->rows_off = base;
    J->vals_off = base + rows_words;
    J->sol_off = J->vals_off + (int64_t)lay.S * lay.W * EW;
    J->state_off = J->sol_off + lay.sol_words;
    J->log_off = J->state_off + lay.state_words - 2 * PIPAMD_DETLOG * EW;
    J->nlog = 0;
    J->nvar = lay.nvar;
    J->nparm = lay.nparm;
    J->ni = lay.ni;
    J->bigparm = lay.bigparm;
    J->tflags = lay.tflags | PIPAMD_T_SORT | (defer ? PIPAMD_T_FRESHROWS : 0);
    J->src_rows = defer ? (int64_t)(uintptr_t)src : 0;
    J->home_sol_off = 0;
    J->L = lay.L;
    J->S = lay.S;
    J->W = lay.W;
    J->status = PIPAMD_ST_RUN;
    J->aux = 0;
    J->npiv = 0;
    J->ncut = 0;
    J->nupd = 0;
    J->ldet = 1;
    for (int i = 0; i < 2 * PIPAMD_MAXDET; i++) J->det[i] = 0;
    J->det[0] = 1;
    J->maxabs = 0;
    J->state_nch = 0;
    J->ebits = ET<T>::BITS;
  }
}

template <class T>
__global__ void pip_batch_results_kernel(const PipJob *jobs, const i64 *arena, int njobs, int nvar, int nparm,
                                         int *status, int *pivots, int *cuts, T *sol_num, T *sol_den) {
  const int b = blockIdx.x;
  const PipJob *J = &jobs[b];
  if (threadIdx.x == 0) {
    if (status) status[b] = J->status;
    if (pivots) pivots[b] = J->npiv;
    if (cuts) cuts[b] = J->ncut;
  }
  const int nn = nvar * (nparm + 1);
  const T *sn = (const T *)(arena + J->sol_off);
  const bool ok = J->status == PIPAMD_ST_SOLUTION;
  if (sol_num)
    for (int e = threadIdx.x; e < nn; e += blockDim.x) sol_num[(size_t)b * nn + e] = ok ? sn[e] : (T)0;
  if (sol_den)
    for (int i = threadIdx.x; i < nvar; i += blockDim.x) sol_den[(size_t)b * nvar + i] = ok ? sn[nn + i] : (T)0;
}

// totals over a batch: [0] pivots [1] cuts [2] rows rewritten [3] jobs finished (solution or nil)
__global__ void pip_batch_counters_kernel(const PipJob *jobs, int njobs, unsigned long long *out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= njobs) return;
  const PipJob *J = &jobs[b];
  atomicAdd(&out[0], (unsigned long long)J->npiv);
  atomicAdd(&out[1], (unsigned long long)J->ncut);
  atomicAdd(&out[2], (unsigned long long)J->nupd);
  if (J->status == PIPAMD_ST_SOLUTION || J->status == PIPAMD_ST_NIL) atomicAdd(&out[3], 1ull);
}

// ---------------------------------------------------------------- expanser for the batch layer
// traiter.c:55-88 / integrer.c:410-415: a tableau whose spare rows are spent is copied into a larger one.  One
// workgroup per entry of the launch list: a job still PIPAMD_ST_RUN is passed on as it is; a job at
// PIPAMD_ST_CAPACITY gets block number atomicAdd(side_count) of the side arena (layout `nl`: same columns, more
// rows; nl.arena_off = the arena-relative offset of the side arena's first block), its row tables and rows are
// copied, the new spare rows zeroed, and it is passed on as PIPAMD_ST_RUN (summaries are rebuilt by the next
// launch, as after the host tree's grow()).  Its determinant log is empty at this point (replayed after every
// launch) and the limbs live in the PipJob.
__global__ __launch_bounds__(256) void pip_rehouse_kernel(PipJob *jobs, i64 *arena, PipQueue q, PipBatchLayout nl,
                                                          int *side_count, int side_cap) {
  const int nq = *q.in_count;
  if ((int)blockIdx.x >= nq) return;
  const int jb = q.in_list[blockIdx.x], tid = threadIdx.x;
  PipJob *J = &jobs[jb];
  const int st = J->status;
  if (st == PIPAMD_ST_RUN) {
    if (tid == 0) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, J->ni);
    }
    return;
  }
  if (st != PIPAMD_ST_CAPACITY || nl.S <= J->S || nl.W != J->W) return;  // cannot grow: the status stands
  __shared__ int s_idx;
  if (tid == 0) s_idx = atomicAdd(side_count, 1);
  __syncthreads();
  if (s_idx >= side_cap) return;
  const int EW = J->ebits == 128 ? 2 : 1;
  const int oL = J->L, oS = J->S, W = J->W, nligne = J->nvar + J->ni;
  const i64 base = nl.arena_off + (i64)s_idx * nl.per_job;
  const i64 rows_words = (i64)nl.L * EW + nl.L;
  const i64 *oden = arena + J->rows_off;
  const int *oflag = (const int *)(oden + (i64)oL * EW), *oref = oflag + oL;
  i64 *nden = arena + base;
  int *nflag = (int *)(nden + (i64)nl.L * EW), *nref = nflag + nl.L;
  for (int e = tid; e < nl.L * EW; e += blockDim.x) nden[e] = e < nligne * EW ? oden[e] : 0;
  for (int k = tid; k < nl.L; k += blockDim.x) {
    nflag[k] = k < nligne ? oflag[k] : 0;
    nref[k] = k < nligne ? oref[k] : 0;
  }
  const i64 *ovals = arena + J->vals_off;
  i64 *nvals = arena + base + rows_words;
  const i64 ow = (i64)oS * W * EW, nw = (i64)nl.S * W * EW;
  for (i64 e = tid; e < nw; e += blockDim.x) nvals[e] = e < ow ? ovals[e] : 0;
  __syncthreads();
  if (tid == 0) {
    if (J->home_sol_off == 0) J->home_sol_off = J->sol_off;
    J->rows_off = base;
    J->vals_off = base + rows_words;
    J->sol_off = J->vals_off + nw;
    J->state_off = J->sol_off + nl.sol_words;
    J->log_off = J->state_off + nl.state_words - 2 * PIPAMD_DETLOG * EW;
    J->L = nl.L;
    J->S = nl.S;
    J->tflags &= ~PIPAMD_T_STATE;
    J->status = PIPAMD_ST_RUN;
    q.out_list[atomicAdd(q.out_count, 1)] = jb;
    atomicMax(q.out_maxni, J->ni);
  }
}
// when the solve ends: the solution of a re-housed job goes back into its own block of the caller's workspace
// (the side arena belongs to the engine and serves the next solve)
__global__ void pip_rehouse_finish_kernel(PipJob *jobs, i64 *arena, int njobs, int sol_words) {
  const int b = blockIdx.x;
  if (b >= njobs) return;
  PipJob *J = &jobs[b];
  const i64 home = J->home_sol_off;
  if (home == 0) return;
  const i64 *src = arena + J->sol_off;
  i64 *dst = arena + home;
  for (int e = threadIdx.x; e < sol_words; e += blockDim.x) dst[e] = src[e];
  __syncthreads();
  if (threadIdx.x == 0) {
    J->sol_off = home;
    J->home_sol_off = 0;
  }
}
extern "C" hipError_t pipk_launch_rehouse(PipJob *jobs, i64 *arena, void *const *q5, int grid, PipBatchLayout nl,
                                          int *side_count, int side_cap, hipStream_t stream) {
  if (grid <= 0) return hipSuccess;
  const PipQueue q{(const int *)q5[0], (const int *)q5[1], (int *)q5[2], (int *)q5[3], (int *)q5[4]};
  hipLaunchKernelGGL(pip_rehouse_kernel, dim3(grid), dim3(256), 0, stream, jobs, arena, q, nl, side_count, side_cap);
  return hipGetLastError();
}
extern "C" hipError_t pipk_launch_rehouse_finish(PipJob *jobs, i64 *arena, int njobs, int sol_words, hipStream_t stream) {
  if (njobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(pip_rehouse_finish_kernel, dim3(njobs), dim3(64), 0, stream, jobs, arena, njobs, sol_words);
  return hipGetLastError();
}

// ---------------------------------------------------------------- forest helpers
// Batched host<->device traffic of the lock-step decision-tree scheduler (pip_forest.cpp):
// one clone pass, one patch pass, one advance launch and one gather pass per step serve every
// problem of the batch.
// clone: list of (src word, dst word, n words) int64 triples -- expanser for a tree split
__global__ void pip_clone_kernel(i64 *arena, const i64 *list, int n) {
  const int b = blockIdx.x;
  if (b >= n) return;
  const i64 *src = arena + list[3 * b];
  i64 *dst = arena + list[3 * b + 1];
  const i64 nw = list[3 * b + 2];
  for (i64 i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
}
// patch: patch p = { dst (32-bit word index into the arena), n, payload[n] } at buf[index[p]]
__global__ void pip_patch_kernel(int *arena32, const int *buf, const i64 *index, int n) {
  const int b = blockIdx.x;
  if (b >= n) return;
  const int *p = buf + index[b];
  const i64 dst = ((i64)(unsigned)p[0]) | ((i64)p[1] << 32);
  const int nw = p[2];
  for (int i = threadIdx.x; i < nw; i += blockDim.x) arena32[dst + i] = p[3 + i];
}
// fresh: build new tableaux (tab_alloc + tab_get, tab.c:158-248) from their rows alone.
// Record r at buf[index[r]] (int64 words): rows_off, nvar, ni, ncol, L, S, W, 0, then ni*ncol values of the entry type T
// (the offsets are the job's, in int64 words; a 128-bit value is two words, low first).
template <class T>
__global__ void pip_fresh_kernel(i64 *arena, const i64 *buf, const i64 *index, int n) {
  constexpr int EW = sizeof(T) / 8;
  const int b = blockIdx.x;
  if (b >= n) return;
  const i64 *p = buf + index[b];
  const i64 rows_off = p[0];
  const int nvar = (int)p[1], ni = (int)p[2], ncol = (int)p[3], L = (int)p[4], S = (int)p[5], W = (int)p[6];
  const T *src = (const T *)(p + 8);
  T *g_den = (T *)(arena + rows_off);
  int *g_flag = (int *)(g_den + L);
  int *g_ref = g_flag + L;
  T *vals = (T *)(arena + rows_off + (i64)L * EW + L);
  for (int i = threadIdx.x; i < nvar + ni; i += blockDim.x) {
    g_den[i] = 1;
    g_flag[i] = i < nvar ? PIPAMD_F_UNIT : PIPAMD_F_UNKNOWN;
    g_ref[i] = i < nvar ? i : i - nvar;
  }
  for (int e = threadIdx.x; e < ni * W; e += blockDim.x) {
    const int s = e / W, j = e % W;
    vals[e] = j < ncol ? src[(size_t)s * ncol + j] : (T)0;
  }
  const int pad = W - ncol;
  for (int e = threadIdx.x; e < (S - ni) * pad; e += blockDim.x) {
    const int s = ni + e / pad, j = ncol + e % pad;
    vals[(size_t)s * W + j] = 0;
  }
}

// gather: what the host needs from each job of the last launch, by status, into out + off[b] (off in int64 words; every
// item is a value of the entry type T):
//   NEED_COMPA  : n, then per undecided row (ascending): row, critic, constant, nparm parameter coefs
//   NEED_PARMCUT: row (aux), denominator, ncol entries
//   SOLUTION    : the solution block (nvar*(nparm+1) numerators, nvar denominators)
template <class T>
__global__ void pip_gather_kernel(const PipJob *jobs, const i64 *arena, int njobs, i64 *out, const i64 *off) {
  const int b = blockIdx.x;
  if (b >= njobs) return;
  const PipJob *J = &jobs[b];
  if (off[b + 1] == off[b]) return;  // the host does not want anything from this job
  T *o = (T *)(out + off[b]);
  const int nvar = J->nvar, nparm = J->nparm, L = J->L, W = J->W, ncol = nvar + nparm + 1;
  const T *g_den = (const T *)(arena + J->rows_off);
  const int *g_flag = (const int *)(g_den + L);
  const int *g_ref = g_flag + L;
  const T *vals = (const T *)(arena + J->vals_off);
  const int lane = threadIdx.x;  // one wave
  if (J->status == PIPAMD_ST_NEED_COMPA) {
    const int nligne = nvar + J->ni;
    const int rec = 3 + nparm;
    int base = 0;
    for (int k0 = 0; k0 < nligne; k0 += 64) {
      const int k = k0 + lane;
      const bool und = k < nligne && (g_flag[k] & (PIPAMD_F_CRITIC | PIPAMD_F_UNKNOWN));
      const u64 m = __ballot(und);
      if (und) {
        const T *r = vals + (size_t)g_ref[k] * W;
        T *q = o + 1 + (size_t)(base + __popcll(m & ((1ull << lane) - 1))) * rec;
        int critic = 1;
        for (int j = 0; j < nvar; j++)
          if (r[j] > 0) {
            critic = 0;
            break;
          }
        q[0] = k;
        q[1] = critic;
        q[2] = r[nvar];
        for (int j = 0; j < nparm; j++) q[3 + j] = r[nvar + 1 + j];
      }
      base += __popcll(m);
    }
    if (lane == 0) o[0] = base;
  } else if (J->status == PIPAMD_ST_NEED_PARMCUT) {
    const int ci = J->aux;
    const T *r = vals + (size_t)g_ref[ci] * W;
    if (lane == 0) {
      o[0] = ci;
      o[1] = g_den[ci];
    }
    for (int j = lane; j < ncol; j += 64) o[2 + j] = r[j];
  } else if (J->status == PIPAMD_ST_SOLUTION) {
    const T *sn = (const T *)(arena + J->sol_off);
    const int n = nvar * (nparm + 1) + nvar;
    for (int e = lane; e < n; e += 64) o[e] = sn[e];
  }
}

extern "C" hipError_t pipk_launch_clone(i64 *arena, const i64 *list, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pip_clone_kernel, dim3(n), dim3(256), 0, stream, arena, list, n);
  return hipGetLastError();
}
extern "C" hipError_t pipk_launch_patch(i64 *arena, const int *buf, const i64 *index, int n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pip_patch_kernel, dim3(n), dim3(128), 0, stream, (int *)arena, buf, index, n);
  return hipGetLastError();
}
extern "C" hipError_t pipk_launch_fresh(i64 *arena, const i64 *buf, const i64 *index, int n, int ebits, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (ebits == 128)
    hipLaunchKernelGGL(pip_fresh_kernel<i128>, dim3(n), dim3(128), 0, stream, arena, buf, index, n);
  else
    hipLaunchKernelGGL(pip_fresh_kernel<i64>, dim3(n), dim3(128), 0, stream, arena, buf, index, n);
  return hipGetLastError();
}
extern "C" hipError_t pipk_launch_gather(const PipJob *jobs, const i64 *arena, int njobs, i64 *out, const i64 *off, int ebits,
                                         hipStream_t stream) {
  if (njobs <= 0) return hipSuccess;
  if (ebits == 128)
    hipLaunchKernelGGL(pip_gather_kernel<i128>, dim3(njobs), dim3(64), 0, stream, jobs, arena, njobs, out, off);
  else
    hipLaunchKernelGGL(pip_gather_kernel<i64>, dim3(njobs), dim3(64), 0, stream, jobs, arena, njobs, out, off);
  return hipGetLastError();
}

// ------------------------------------------------------------------ launchers
// columns a wave's registers cover (row chunks x 16 B per lane), by entry width
static int wp_of(int Wmax, int ebits) {
  if (ebits == 128) return Wmax <= 64 ? 64 : (Wmax <= 128 ? 128 : (Wmax <= 256 ? 256 : 512));
  return Wmax <= 128 ? 128 : (Wmax <= 256 ? 256 : 512);
}

extern "C" size_t pipk_advance_lds_bytes(int Lmax, int Smax, int Wmax, int ebits) {
  const size_t WP = (size_t)wp_of(Wmax, ebits);
  const size_t NM = WP / 64, EB = ebits == 128 ? 16 : 8;
  size_t shm = EB * 2 * (size_t)Smax + prow_bytes(EB * WP, Smax) + sizeof(u64) * (size_t)Smax * NM +
               sizeof(u16) * (3 * (size_t)Smax + (size_t)Lmax + WP) + 3 * (size_t)Smax;
  return (shm + 15) & ~(size_t)15;
}

extern "C" hipError_t pipk_launch_batch_counters(const PipJob *jobs, int njobs, unsigned long long *out,
                                                 hipStream_t stream) {
  hipError_t e = hipMemsetAsync(out, 0, 4 * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(pip_batch_counters_kernel, dim3((njobs + 255) / 256), dim3(256), 0, stream, jobs, njobs, out);
  return hipGetLastError();
}

// the one-wave kernel of <= 128 int64 columns with a compile-time row capacity (see SC above)
template <int SC>
static hipError_t launch_static(AdvanceLaunch a, int ebits) {
  a.Smax = SC;
  a.Lmax = SC + 128;
  a.shm = pipk_advance_lds_bytes(a.Lmax, a.Smax, 128, ebits);
  return a.full ? launch_advance_t<i64, 1, 1, false, SC, true>(a) : launch_advance_t<i64, 1, 1, false, SC, false>(a);
}
// Row-capacity class of the one-wave kernels with a compile-time LDS image for a launch of `smax` row slots: the
// smallest class that holds it if that costs at most an eighth more LDS than its exact size (occupancy is LDS-bound
// at 24 tableaux per CU); 0 = none.
extern "C" int pipk_static_class(int smax) {
  const int s = (smax + 3) & ~3;
  if (s <= 64) return 64;
  if (s > 84 && s <= 96) return 96;
  if (s > 98 && s <= 112) return 112;
  if (s > 112 && s <= 128) return 128;
  if (s > 140 && s <= 160) return 160;
  return 0;
}
// Row-capacity class of the lean kernel (pip_lean.h) for `smax` row slots: the smallest that holds them (a class too
// large costs a little occupancy, no class costs the lean launch); 0 = none.
extern "C" int pipk_lean_class(int smax) {
  const int s = (smax + 3) & ~3;
  return s <= 64 ? 64 : (s <= 96 ? 96 : (s <= 112 ? 112 : (s <= 128 ? 128 : (s <= 160 ? 160 : 0))));
}
extern "C" size_t pipk_lean64_lds_bytes(int Smax, int Lmax) { return lean64_lds_bytes((Smax + 3) & ~3, (Lmax + 3) & ~3); }
// the lean bulk kernel over a launch list; the caller has checked pipk_lean_class(a.Smax) != 0
static hipError_t launch_lean_class(AdvanceLaunch a) {
  const int sc = pipk_lean_class(a.Smax);
  a.Smax = sc;
  a.Lmax = sc + 128;
  a.shm = pipk_advance_lds_bytes(a.Lmax, a.Smax, 128, 64);
  switch (sc) {
    case 64: return a.full ? launch_lean<64, true>(a) : launch_lean<64, false>(a);
    case 96: return a.full ? launch_lean<96, true>(a) : launch_lean<96, false>(a);
    case 112: return a.full ? launch_lean<112, true>(a) : launch_lean<112, false>(a);
    case 128: return a.full ? launch_lean<128, true>(a) : launch_lean<128, false>(a);
    case 160: return a.full ? launch_lean<160, true>(a) : launch_lean<160, false>(a);
  }
  return hipErrorInvalidValue;
}
template <class T, int NCH>
static hipError_t launch_advance_w(bool one, const AdvanceLaunch &a) {
  if constexpr (sizeof(T) == 8) {
    if (a.gimg) return launch_advance_t<T, NCH, 4, true, 0, false>(a);  // tables in HBM: four waves per job
#if !defined(PIP_NO_STATIC_IMAGE)
    if constexpr (NCH == 1) {
      // row-capacity classes; a launch goes to the smallest class that holds it if that costs at most
      // an eighth more LDS than its exact size (occupancy is LDS-bound at 24 tableaux per CU)
      if (one && a.Lmax - a.Smax <= 128) {
        switch (pipk_static_class(a.Smax)) {
          case 64: return launch_static<64>(a, 64);
          case 96: return launch_static<96>(a, 64);
          case 112: return launch_static<112>(a, 64);
          case 128: return launch_static<128>(a, 64);
          case 160: return launch_static<160>(a, 64);
        }
      }
    }
#endif
  }
  if constexpr (sizeof(T) == 8 && NCH == 1) {
    // eight waves per job: the rows of a late pivot of a long tableau (15-25 of them change) are
    // spread over twice the waves -- for the few tableaux of a tail launch
    if (a.waves == 8) return launch_advance_t<T, NCH, 8, false, 0, false>(a);
  }
  if constexpr (sizeof(T) == 16 && NCH <= 4) {
    // sixteen waves per job, a whole CU: the hundreds of rows a late pivot of a long 128-bit tableau rewrites, spread over
    // four times the waves -- for a tail launch over a few such tableaux (fewer than the GPU has CUs)
    if (a.waves == 16) return launch_advance_t<T, NCH, 16, false, 0, false>(a);
  }
  return one ? launch_advance_t<T, NCH, 1, false, 0, false>(a) : launch_advance_t<T, NCH, 4, false, 0, false>(a);
}

// waves_per_job: 1 = one wave64 per tableau (latency-bound sparse batches: more tableaux in
// flight per CU), 4 = four waves share a tableau's rows (few, large tableaux).
// ebits: 64 or 128 -- every job of the launch must have that entry width.
static hipError_t launch_by_shape(const AdvanceLaunch &a, bool one, int wp, int ebits);

// q5: NULL = job b is workgroup b's; else five device pointers {in_list, in_count, out_list,
// out_count, out_maxni} (see PipQueue; the in_ pair and the out_ triple may each be NULL) and
// `grid` = an upper bound on *in_count (0: njobs).
// big: NULL, or {void **buffer, size_t *bytes} of the caller -- a device buffer this function
// (re)allocates when the row tables of the launch do not fit LDS (64-bit entries only): the launch
// then keeps them there, `grid` blocks of the image size.  Without it such a launch is refused.
// hints: bit 0 = every job of the launch has no parameters, no big parameter and nvar + 1 == W ==
// the wave's column coverage (the caller knows its batch is uniform): see FULL.  Bit 1 (one wave per job, 64-bit entries, at
// most 128 columns and pipk_lean_class(Smax) != 0, else refused) = the lean kernel of pip_lean.h: it runs the jobs it
// can (no parameters, entries below 2^15) and leaves the others PIPAMD_ST_RUN on the output list for a launch without
// this bit.  Bit 2: no determinant replay behind the launch (see pipk_launch_replay_all).  Bit 3 (one wave per job, 128-bit
// entries, 129 ... 256 columns, pipk_lean64_lds_bytes(Smax, Lmax) within the LDS budget, else refused) = the lean kernel of
// pip_lean64.h: it runs the jobs it can (no parameters, entries below 2^63) and leaves the others on the output list.
extern "C" hipError_t pipk_launch_advance_q(PipJob *jobs, i64 *arena, int njobs, int Lmax, int Smax, int Wmax,
                                            int iter_limit, int waves_per_job, int ebits, void *const *q5, int grid,
                                            void **big, int hints, unsigned long long *prof, hipStream_t stream) {
  if (njobs <= 0) return hipSuccess;
  // LDS arrays are carved at 16/8/4/2/1-byte granularity in that order: keep Lmax, Smax multiples of 4
  Lmax = (Lmax + 3) & ~3;
  Smax = (Smax + 3) & ~3;
  if (Wmax > 512) return hipErrorInvalidValue;
  AdvanceLaunch a;
  a.jobs = jobs;
  a.arena = arena;
  a.njobs = njobs;
  a.Lmax = Lmax;
  a.Smax = Smax;
  a.Wmax = Wmax;
  a.iter_limit = iter_limit;
  a.q = PipQueue{nullptr, nullptr, nullptr, nullptr, nullptr};
  a.grid = njobs;
  if (q5) {
    a.q = PipQueue{(const int *)q5[0], (const int *)q5[1], (int *)q5[2], (int *)q5[3], (int *)q5[4]};
    a.grid = grid;
  }
  a.prof = prof;
  a.waves = waves_per_job;
  a.full = (hints & 1) != 0;
  a.shm = pipk_advance_lds_bytes(Lmax, Smax, Wmax, ebits);
  a.gimg = nullptr;
  a.gslots = 0;
  if (a.shm > PIPAMD_LDS_BUDGET) {  // the row tables of this job mix do not fit a CU's LDS
    if (ebits != 64 || !big) return hipErrorInvalidConfiguration;
    void **buf = (void **)big[0];
    size_t *cap = (size_t *)big[1];
    // A pool of image blocks, not one per workgroup: a launch over a 10k-tableau list of which a handful are
    // unfinished would otherwise pin (and possibly fail to get) gigabytes.  Workgroup b uses block b % slots
    // behind a lock word; 1,024 blocks cover the workgroups a GPU keeps resident (256 CUs x at most 4 of these
    // 256-thread, ~100-register workgroups), so a workgroup seldom waits.
    const int wgs = a.grid > 0 && a.grid < njobs ? a.grid : njobs;
    a.gslots = wgs < 1024 ? wgs : 1024;
    const size_t locks = ((size_t)a.gslots * sizeof(int) + 255) & ~(size_t)255;
    const size_t need = locks + (size_t)a.gslots * a.shm;
    if (*cap < need) {
      if (*buf) {
        hipError_t fe = hipFree(*buf);  // waits for the launches that may still use it
        if (fe != hipSuccess) return fe;
      }
      *buf = nullptr;
      *cap = 0;
      hipError_t me = hipMalloc(buf, need);
      if (me != hipSuccess) return me;
      *cap = need;
      me = hipMemsetAsync(*buf, 0, need, stream);  // every lock free; re-zeroed below for a smaller pool
      if (me != hipSuccess) return me;
    }
    a.gimg = (unsigned char *)*buf;
    // the lock words sit at the head of the buffer whatever the pool size: all are free between launches
    // (every workgroup releases its block before it ends), so nothing needs zeroing per launch
  }
  a.stream = stream;
  const bool one = waves_per_job == 1;
  const int wp = wp_of(Wmax, ebits);
  hipError_t le;
  if (hints & 2) {
    if (!one || ebits != 64 || wp != 128 || a.gimg || !pipk_lean_class(a.Smax)) return hipErrorInvalidValue;
    le = launch_lean_class(a);
  } else if (hints & 8) {  // the lean kernel of the 128-bit flavour (pip_lean64.h): 129 ... 256 columns, one wave per job
    if (!one || ebits != 128 || wp != 256 || pipk_lean64_lds_bytes(a.Smax, a.Lmax) > PIPAMD_LDS_BUDGET) return hipErrorInvalidValue;
    le = launch_lean64(a);
  } else {
    le = launch_by_shape(a, one, wp, ebits);
  }
  if (le != hipSuccess) return le;
  // the determinant bookkeeping of the pivots just logged (hints bit 2: the caller replays later -- pipk_launch_replay_all)
  if (hints & 4) return hipGetLastError();
  const int nrep = a.grid > 0 && a.grid < njobs ? a.grid : njobs;
  if (one) {  // behind a bulk launch: fewest instructions
    if (ebits == 128)
      hipLaunchKernelGGL(pip_det_replay_lanes_kernel<i128>, dim3((nrep + 63) / 64), dim3(64), 0, stream, jobs, arena, njobs, a.q);
    else
      hipLaunchKernelGGL(pip_det_replay_lanes_kernel<i64>, dim3((nrep + 63) / 64), dim3(64), 0, stream, jobs, arena, njobs, a.q);
  } else {  // few jobs, someone is waiting for them: shortest latency
    if (ebits == 128)
      hipLaunchKernelGGL(pip_det_replay_kernel<i128>, dim3(nrep), dim3(64), 0, stream, jobs, arena, njobs, a.q);
    else
      hipLaunchKernelGGL(pip_det_replay_kernel<i64>, dim3(nrep), dim3(64), 0, stream, jobs, arena, njobs, a.q);
  }
  return hipGetLastError();
}

// The determinant logs of ALL jobs 0..njobs-1 replayed: behind a sequence of launches that ran with hints bit 2.
// wave_per_job = 0: one lane per job (jobs with an empty log cost a load) -- fewest instructions, what counts when other
// batches keep the device busy; 1: one wave per job -- shortest latency (a lane walks its up to 200 log entries alone for
// half a millisecond), for a caller that runs one batch at a time.  The log holds PIPAMD_DETLOG pivots and the pivot
// kernels pause a job whose log is full, so a sequence may log at most that many pivots per job between replays.
extern "C" hipError_t pipk_launch_replay_all(PipJob *jobs, i64 *arena, int njobs, int ebits, int wave_per_job, hipStream_t stream) {
  if (njobs <= 0) return hipSuccess;
  const PipQueue q{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (wave_per_job) {
    if (ebits == 128)
      hipLaunchKernelGGL(pip_det_replay_kernel<i128>, dim3(njobs), dim3(64), 0, stream, jobs, arena, njobs, q);
    else
      hipLaunchKernelGGL(pip_det_replay_kernel<i64>, dim3(njobs), dim3(64), 0, stream, jobs, arena, njobs, q);
  } else if (ebits == 128)
    hipLaunchKernelGGL(pip_det_replay_lanes_kernel<i128>, dim3((njobs + 63) / 64), dim3(64), 0, stream, jobs, arena, njobs, q);
  else
    hipLaunchKernelGGL(pip_det_replay_lanes_kernel<i64>, dim3((njobs + 63) / 64), dim3(64), 0, stream, jobs, arena, njobs, q);
  return hipGetLastError();
}

static hipError_t launch_by_shape(const AdvanceLaunch &a, bool one, int wp, int ebits) {
#ifdef PIP_ONLY_MAIN  // diagnostic builds (tools/isa_lines.sh): only the 64-bit, <= 128-column, one-wave kernel
  return (ebits == 64 && wp == 128 && one && !a.gimg) ? launch_advance_w<i64, 1>(true, a) : hipErrorInvalidValue;
#else
  if (ebits == 128) {
    switch (wp) {
      case 64: return launch_advance_w<i128, 1>(one, a);
      case 128: return launch_advance_w<i128, 2>(one, a);
      case 256: return launch_advance_w<i128, 4>(one, a);
      default: return launch_advance_w<i128, 8>(one, a);
    }
  }
  switch (wp) {
    case 128: return launch_advance_w<i64, 1>(one, a);
    case 256: return launch_advance_w<i64, 2>(one, a);
    default: return launch_advance_w<i64, 4>(one, a);
  }
#endif
}

extern "C" hipError_t pipk_launch_advance(PipJob *jobs, i64 *arena, int njobs, int Lmax, int Smax, int Wmax,
                                          int iter_limit, int waves_per_job, int ebits, unsigned long long *prof,
                                          hipStream_t stream) {
  return pipk_launch_advance_q(jobs, arena, njobs, Lmax, Smax, Wmax, iter_limit, waves_per_job, ebits, nullptr, 0, nullptr,
                               0, prof, stream);
}

extern "C" hipError_t pipk_launch_batch_load(PipJob *jobs, i64 *arena, const i64 *rows, PipBatchLayout lay, int first,
                                             int count, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  if (lay.ebits == 128)
    hipLaunchKernelGGL(pip_batch_load_kernel<i128>, dim3(count), dim3(256), 0, stream, jobs, arena, rows, lay, first);
  else
    hipLaunchKernelGGL(pip_batch_load_kernel<i64>, dim3(count), dim3(256), 0, stream, jobs, arena, rows, lay, first);
  return hipGetLastError();
}

extern "C" hipError_t pipk_launch_batch_results(const PipJob *jobs, const i64 *arena, int njobs, int nvar, int nparm,
                                                int ebits, int *status, int *pivots, int *cuts, void *sol_num,
                                                void *sol_den, hipStream_t stream) {
  if (ebits == 128)
    hipLaunchKernelGGL(pip_batch_results_kernel<i128>, dim3(njobs), dim3(256), 0, stream, jobs, arena, njobs, nvar, nparm,
                       status, pivots, cuts, (i128 *)sol_num, (i128 *)sol_den);
  else
    hipLaunchKernelGGL(pip_batch_results_kernel<i64>, dim3(njobs), dim3(256), 0, stream, jobs, arena, njobs, nvar, nparm,
                       status, pivots, cuts, (i64 *)sol_num, (i64 *)sol_den);
  return hipGetLastError();
}
