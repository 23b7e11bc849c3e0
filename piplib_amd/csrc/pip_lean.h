// piplib_amd/csrc/pip_lean.h -- the lean bulk kernel: pip_advance_kernel's pivot loop specialised for the regime the
// headline workload lives in.
//
// One wave per tableau, at most 127 unknowns + constant in rows of W <= 128 columns (FULL: exactly 127 + 1, compile-time
// column counts), no parameters, no big parameter, 64-bit Entier, compile-time row capacity SC, rows skipped, plain
// cuts -- and EVERY entry of EVERY row below 2^31 in magnitude, i.e. an int.  Under that invariant
//   * rows live in HBM as int32 (4 W bytes per row, in the first half of the row's slot of W long longs): half the
//     traffic of the reference's long long rows, half the working set; a row is two 32-bit registers per lane;
//   * while the rows involved are in magnitude class 0 (entries below 2^15; pivot row's denominator too) every product of
//     a pivot fits 31 bits: the elimination is 24-bit multiplies, the row gcd float-reciprocal remainders, the exact
//     division, the summaries, choisir_piv's cross products (24-bit) and the cuts 32-bit arithmetic (the "small" path,
//     96 % of the headline's pivots);
//   * a row of class 1 (an entry between 2^15 and 2^31), or a pivot row of class 1, takes the "mid" path (round 4): the same
//     int rows, products in 64-bit registers (v_mad_i64_i32; below 2^62 each, so nothing wraps), the row gcd and the
//     division through row_reduce<i64> -- the code pip_advance_kernel runs on such a row, on the same values -- and
//     choisir_piv's cross products in 64 bits.  The result is an int row again, almost always;
//   * none of the general kernel's other paths (wide tournament, parameters, deepest cuts, row tables in HBM) is
//     compiled in, the pivot row stays in registers, the LDS image is smaller (lean_lds_bytes): 64 VGPRs, no
//     scratch, eight waves per SIMD.
// A rewritten row that does NOT fit ints any more is stored in the general format (W long longs, the whole slot) and
// summarised with pip_advance_kernel's magnitude classes (2, 3); the pivot is finished -- no row is read twice in a pivot
// -- and the running maximum of the classes, checked between pivots, then ends the lean run.  A tableau that leaves --
// or anything else this kernel does not do: entries beyond ints at entry, a cut under a denominator of 2^31 or more,
// PIPAMD_T_NOSKIP / _DEEPEST, a paused job -- is handed over in the general format (int rows widened to int64 in
// place, the same row tables and saved summaries as a paused job of pip_advance_kernel) and stays PIPAMD_ST_RUN on the
// launch list: pipamd_batch_solve's next launches (pip_advance_kernel) take it from there.  Same algorithm, same
// statuses, same bits as pip_advance_kernel -- the reference's traiter()/pivoter()/choisir_piv()/exam_coef()/integrer()/tab_sort_rows
// (traiter.c:101-159, 297-548, 556-623, 628-791; integrer.c:305-486) -- which the parity tests check tableau by tableau
// (tests/test_gpu_parity.py: test_lean_kernel_paths, test_lean_kernel_other_widths, and every batch test of the suite).
#ifndef PIP_LEAN_H
#define PIP_LEAN_H
#include "pip_advance.h"

#ifndef PIP_LEAN_PF
#define PIP_LEAN_PF 2  // rows of a pivot's work list in flight
#endif
#ifndef PIP_LEAN_MID_INV
#define PIP_LEAN_MID_INV 0  // (A/B switch) the mid path's row gcd and division by inverse multiplication (12 bytes of scratch per lane)
#endif
#ifndef PIP_LEAN_WAVES
#define PIP_LEAN_WAVES 8  // waves per SIMD the kernel is bounded to (64 VGPRs)
#endif

// a packed row: lane l holds columns 2l, 2l+1 as ints, 8 bytes per lane
// (W: the columns of a row, even and <= 128; lanes beyond them hold zeros)
__device__ __forceinline__ void row_load32p(RowRegs32<1> &r, const i64 *slot, int lane, int W = 128) {
  int2 t = {0, 0};
  if (2 * lane < W) t = *reinterpret_cast<const int2 *>(reinterpret_cast<const int *>(slot) + 2 * lane);
  r.v[0][0] = t.x;
  r.v[0][1] = t.y;
}
__device__ __forceinline__ void row_store32p(const RowRegs32<1> &r, i64 *slot, int lane, int W = 128) {
  int2 t;
  t.x = r.v[0][0];
  t.y = r.v[0][1];
  if (2 * lane < W) *reinterpret_cast<int2 *>(reinterpret_cast<int *>(slot) + 2 * lane) = t;
}

// a row that no longer fits ints: the general format, the whole slot
__device__ __forceinline__ void row_store64w(const i64 (&z)[2], i64 *slot, int lane, int W = 128) {
  longlong2 t;
  t.x = z[0];
  t.y = z[1];
  if (2 * lane < W) *reinterpret_cast<longlong2 *>(slot + 2 * lane) = t;
}

// rows [0, n) of a block, packed -> the general format, each within its own slot (the loads of a group of rows are back
// before their slots are overwritten); rows of class 2 or 3 (rcls, LDS) are in the general format already
__device__ __forceinline__ void rows_unpack(i64 *vals, int n, int lane, int W, const u8 *rcls) {
  for (int s0 = 0; s0 < n; s0 += 4) {
    RowRegs32<1> rr[4];
    bool packed[4];
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      packed[qq] = s0 + qq < n && rcls[s0 + qq] < 2;
      if (packed[qq]) row_load32p(rr[qq], vals + (size_t)(s0 + qq) * W, lane, W);
    }
#pragma unroll
    for (int qq = 0; qq < 4; qq++)
      if (packed[qq] && 2 * lane < W) {
        longlong2 t;
        t.x = (i64)rr[qq].v[0][0];
        t.y = (i64)rr[qq].v[0][1];
        *reinterpret_cast<longlong2 *>(vals + (size_t)(s0 + qq) * W + 2 * lane) = t;
      }
  }
}

// row_publish32<1> for this kernel's LDS image (constant terms kept as ints): sign summary, non-zero bitmap and
// magnitude class of a row of nvar unknowns + constant; returns the class (0: every entry below 2^15)
__device__ __forceinline__ int lean_publish(const RowRegs32<1> &z, const Shared<i64> &S, int *cst, int s, int pivj, int extra_sig,
                                            int lane, int nvar = 127) {
  const int cz = row_entry32<1>(z, 0, nvar & 1, nvar >> 1);  // the constant term, column nvar
  int sig = extra_sig | (cz > 0 ? 1 : (cz < 0 ? 2 : 0));
  if (pivj >= 0) {
    const int pz = row_entry32<1>(z, 0, pivj & 1, pivj >> 1);
    sig |= (pz > 0 ? 1 : (pz < 0 ? 2 : 0)) << 6;
  }
  const int v0 = z.v[0][0], v1 = z.v[0][1];
  const unsigned mx = (unsigned)(v0 < 0 ? -v0 : v0) | (unsigned)(v1 < 0 ? -v1 : v1);
  const u64 nz0 = ballot64(v0 != 0), nz1 = ballot64(v1 != 0);
  const int cls = ballot64((mx >> 15) != 0) ? 1 : 0;
  if (lane == 0) {
    S.sig[s] = (u16)sig;
    S.rcls[s] = (u8)cls;
    cst[s] = cz;
    S.nzm[(size_t)s * 2] = nz0;
    S.nzm[(size_t)s * 2 + 1] = nz1;
  }
  return cls;
}

// the same for a row that left the ints (z: lane l's columns 2l, 2l+1 as long longs): pip_advance_kernel's classes
// (2: below 2^47, 3: beyond; 1 only for an entry of exactly -2^31...); the constant term kept here is truncated -- the
// lean run ends before anything reads it
__device__ __forceinline__ int lean_publish_wide(const i64 (&z)[2], const Shared<i64> &S, int *cst, int s, int pivj, int extra_sig,
                                                 int lane, int nvar = 127) {
  const i64 cz = readlane64((nvar & 1) ? z[1] : z[0], nvar >> 1);
  int sig = extra_sig | sign_code(cz);
  if (pivj >= 0) sig |= sign_code(readlane64((pivj & 1) ? z[1] : z[0], pivj >> 1)) << 6;
  const u64 nz0 = ballot64(z[0] != 0), nz1 = ballot64(z[1] != 0);
  int cls = cls_of<i64>(uabs64(z[0]) | uabs64(z[1]));
  if (cls < 2) cls = 2;  // (it does not fit an int: at least 2^31)
  if (lane == 0) {
    S.sig[s] = (u16)sig;
    S.rcls[s] = (u8)cls;
    cst[s] = (int)cz;
    S.nzm[(size_t)s * 2] = nz0;
    S.nzm[(size_t)s * 2 + 1] = nz1;
  }
  return cls;
}

// bytes of this kernel's LDS image for SC row slots (smaller than pip_advance_kernel's: no pivot row, int constants)
__host__ __device__ constexpr size_t lean_lds_bytes(int SC) { return ((size_t)39 * SC + 2 * 128 + 2 * 128 + 15) & ~(size_t)15; }

// choisir_piv (traiter.c:297-341) as choose_column<i64, 1, SMALL> does it, on packed rows.  SMALL: every row of the
// tableau is in class 0, the cross products are 24-bit multiplies; else entries are below 2^31, the products below 2^62
// and their difference a long long.
template <bool SMALL>
__device__ __forceinline__ int choose_column32(const Shared<i64> &S, const RowRegs32<1> &prow, const i64 *vals, int W, int nvar,
                                               int nligne, int pivi, Scalars *sc) {
  constexpr int NM = 2;
  const int lane = threadIdx.x & 63;
  int a[2], u[2];
  bool cand[2];
  u64 cm[NM];
  int count = 0;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int j = 2 * lane + h;
    a[h] = j < nvar ? prow.v[0][h] : 0;
    cand[h] = a[h] > 0;
    u[h] = cand[h] ? (int)S.urow[j] : -1;
    cm[h] = ballot64(cand[h]);
    count += __popcll(cm[h]);
  }
  if (count == 0) return -1;
  for (int k0 = 0; k0 < nligne && count > 1; k0 += 64) {
    const int k = k0 + lane;
    bool rel = false;
    if (k < nligne && k != pivi) {
      const int rf = S.ref[k];
      if (!(rf & UNITBIT)) {
        const u64 *m = S.nzm + (size_t)rf * NM;
        rel = ((m[0] & cm[0]) | (m[1] & cm[1])) != 0;
      }
    }
    u64 relmask = ballot64(rel);
    while (relmask && count > 1) {
      const int kk = k0 + __ffsll((long long)relmask) - 1;
      relmask &= relmask - 1;
      const int sl = S.ref[kk];
      // unit rows above kk knock out their own column
      int nel = 0;
#pragma unroll
      for (int h = 0; h < 2; h++) nel += __popcll(ballot64(cand[h] && u[h] < kk));
      if (nel == count) goto last_unit_wins;
      if (nel) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          if (u[h] < kk) cand[h] = false;
          cm[h] = ballot64(cand[h]);
        }
        count -= nel;
        if (count == 1) break;
      }
      if (!((S.nzm[(size_t)sl * NM] & cm[0]) | (S.nzm[(size_t)sl * NM + 1] & cm[1]))) continue;  // cannot separate them
      // real row kk: keep the minimal ratios
      RowRegs32<1> n;
      row_load32p(n, vals + (size_t)sl * W, lane, W);
      for (;;) {
        // reference column b = first remaining candidate
        int ab, nb;
        if (cm[0]) {
          const int src = __ffsll((long long)cm[0]) - 1;
          ab = __builtin_amdgcn_readlane(a[0], src);
          nb = __builtin_amdgcn_readlane(n.v[0][0], src);
        } else {
          const int src = __ffsll((long long)cm[1]) - 1;
          ab = __builtin_amdgcn_readlane(a[1], src);
          nb = __builtin_amdgcn_readlane(n.v[0][1], src);
        }
        bool neg[2];
        int nneg = 0, nzero = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          bool xneg, xzero;
          if constexpr (SMALL) {
            const int x = __mul24(ab, n.v[0][h]) - __mul24(nb, a[h]);
            xneg = x < 0;
            xzero = x == 0;
          } else {
            const i64 x = (i64)ab * (i64)n.v[0][h] - (i64)nb * (i64)a[h];
            xneg = x < 0;
            xzero = x == 0;
          }
          neg[h] = cand[h] && xneg;
          const bool zero = cand[h] && xzero;
          nneg += __popcll(ballot64(neg[h]));
          nzero += __popcll(ballot64(zero));
          if (!neg[h] && !zero) cand[h] = false;  // strictly larger: out
        }
        if (nneg == 0) {
          count = nzero;
        } else {
          cand[0] = neg[0];
          cand[1] = neg[1];
          count = nneg;
        }
        cm[0] = ballot64(cand[0]);
        cm[1] = ballot64(cand[1]);
        if (nneg == 0 || count == 1) break;
      }
    }
  }
  if (count == 1) return cm[0] ? 2 * (__ffsll((long long)cm[0]) - 1) : 2 * (__ffsll((long long)cm[1]) - 1) + 1;
last_unit_wins:
  // only unit rows left to look at: the column whose unit row comes last survives
  if (lane == 0) sc->tmp2 = -1;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int h = 0; h < 2; h++)
    if (cand[h]) atomicMax(&sc->tmp2, (u[h] << 10) | (2 * lane + h));
  __builtin_amdgcn_wave_barrier();
  return sc->tmp2 & 1023;
}

// FULL: 127 unknowns + constant, a row fills the wave's 128 columns (the launcher's promise, as for pip_advance_kernel);
// else any number of unknowns up to 127 without parameters, rows of W <= 128 columns (W even).
template <int SC, bool FULL>
__global__ __launch_bounds__(64, PIP_LEAN_WAVES) void pip_lean_kernel(PipJob *jobs, i64 *arena, int njobs, int iter_limit,
                                                                      PipQueue q
#ifdef PIP_PROFILE
                                                                      , u64 *prof
#endif
) {
  typedef i64 T;
  // (diagnostic build only, tools/dbg_prof_lean.py: cycle stamps per piece of the loop -- 0 exam/integrer, 1 pivot row
  // load, 2 choisir_piv, 3 work list, 4 queue + recycled slot, 5 wait for a work row, 6 multipliers, 7 products + row gcd +
  // division, 8 store + summary, 9 phase C, 10 entry, 11 epilogue)
  PROF_DECL;
  constexpr int Smax = SC, Lmax = SC + 128, WP = 128, NM = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ Scalars sc;
  const int nq = q.in_count ? *q.in_count : njobs;
  if ((int)blockIdx.x >= nq) return;
  const int jb = q.in_list ? q.in_list[blockIdx.x] : (int)blockIdx.x;
  PipJob *J = &jobs[jb];
  const int lane = threadIdx.x;
  if (J->status != PIPAMD_ST_RUN) {
    if (J->status == PIPAMD_ST_CAPACITY && q.out_count && lane == 0) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, PIPAMD_Q_CAPFLAG | J->ni);
      atomicAdd(q.out_maxni + 1, 1);  // (the list\'s third control word: tableaux out of rows)
    }
    return;
  }
  int tflags = J->tflags;
  int ni = J->ni;
  const int nvar = FULL ? 127 : J->nvar, W = FULL ? 128 : J->W;
  int nligne = nvar + ni;
  // what this kernel does not do stays with pip_advance_kernel: the job goes on the launch list untouched
  const bool mine = J->nvar == nvar && nvar < 128 && J->nparm == 0 && J->bigparm < 0 && J->W == W && W <= 128 && !(W & 1) &&
                    J->ebits != 128 && !(tflags & (PIPAMD_T_NOSKIP | PIPAMD_T_DEEPEST)) && ni <= Smax && nligne <= Lmax &&
                    (!(tflags & PIPAMD_T_STATE) || J->state_nch == 1);
  if (!mine) {
    if (lane == 0 && q.out_count) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, ni);
    }
    return;
  }
  T *vals = (T *)(arena + J->vals_off);
  // (what only the prologue and the epilogue need -- the row tables in HBM, the saved summaries, the counters -- is
  // derived from the job header where it is used, so that it holds no scalar registers across the pivot loop)
  const int ncut0 = J->ncut - ni;  // cuts so far = ncut0 + ni (every row this kernel appends is a cut)
  const int cap_ni = min(J->S, J->L - nvar);  // rows the job's block holds
  int npiv = J->npiv, nupd = J->nupd;
  T *g_log = (T *)(arena + J->log_off);
  constexpr int LOGCAP = PIPAMD_DETLOG;
  int nlog = J->nlog;

  Shared<T> S;  // the tables of pip_advance_kernel's image this kernel uses
  int *cst;     // [S] constant terms (ints here); the entry-time sort keys share their storage
  {
    unsigned char *p = smem;
    S.den = (T *)p;      p += sizeof(T) * Smax;
    S.nzm = (u64 *)p;    p += sizeof(u64) * (size_t)Smax * NM;
    cst = (int *)p;
    S.size = (float *)p; p += sizeof(int) * Smax;
    S.prow = nullptr;
    S.cst = nullptr;
    S.sig = (u16 *)p;    p += sizeof(u16) * Smax;
    S.srow = (u16 *)p;   p += sizeof(u16) * Smax;
    S.work = (u16 *)p;   p += sizeof(u16) * Smax;
    S.ref = (u16 *)p;    p += sizeof(u16) * Lmax;
    S.urow = (u16 *)p;   p += sizeof(u16) * WP;
    S.fl = (u8 *)p;      p += Smax;
    S.nf = (u8 *)p;      p += Smax;
    S.rcls = (u8 *)p;    p += Smax;
  }

  // ---- the row tables (as pip_advance_kernel stages them)
  for (int j = lane; j < WP; j += 64) S.urow[j] = NOROW;
  if (lane == 0) {
    sc.ovf = 0;
    sc.aux = 0;
    sc.smaxbits = 0;
    sc.pivi = BIG_I;
    sc.pivi2 = BIG_I;
    sc.flagor = 0;
    sc.bad = 0;
  }
  __builtin_amdgcn_wave_barrier();
  {
  const int L = J->L;
  const T *g_den = (const T *)(arena + J->rows_off);
  const int *g_flag = (const int *)(g_den + L);
  const int *g_ref = g_flag + L;
  for (int i = lane; i < nligne; i += 64) {
    const int f = g_flag[i], rf = g_ref[i];
    if (f & PIPAMD_F_UNIT) {
      S.ref[i] = (u16)(UNITBIT | ((f & PIPAMD_F_ZERO) ? UNITZERO : 0) | rf);
      S.urow[rf] = (u16)i;
    } else {
      S.ref[i] = (u16)rf;
      S.srow[rf] = (u16)i;
      S.fl[rf] = (u8)f;
      S.den[rf] = g_den[i];
      S.nf[rf] = 0;
    }
  }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- one pass over the tableau: the rows become ints (rows of a job loaded with PIPAMD_T_ROWS_STAY come from the
  // caller's array), summaries, sort keys.  A row with an entry of 2^31 or more: not a job for this kernel.
  int mcw = 0;  // largest magnitude class published so far: 0 small path everywhere, 1 int rows, beyond: the lean run ends
  {
    constexpr int PF0 = 4;
    // a job that paused in an earlier launch (this kernel's or pip_advance_kernel's, rows in the general format): what the
    // entry pass cannot see in the rows -- "gcd(row, denominator) is known to be 1" -- comes from the saved summaries
    const u16 *g_sig = (tflags & PIPAMD_T_STATE) ? (const u16 *)((const u64 *)(arena + J->state_off) + (size_t)J->S * NM) : nullptr;
    const bool fresh = (tflags & PIPAMD_T_FRESHROWS) != 0;
    const T *src = fresh ? (const T *)(uintptr_t)J->src_rows : vals;
    const int pitch = fresh ? nvar + 1 : W;  // the caller's rows are nvar + 1 wide (an even number: pipamd_batch_load), the block's W
    int npacked = 0;
    bool wide = false;
    for (int s0 = 0; s0 < ni && !wide; s0 += PF0) {
      RowRegs<T, 1> rr[PF0];
#pragma unroll
      for (int qq = 0; qq < PF0; qq++)
        if (s0 + qq < ni) row_load<T, 1>(rr[qq], src + (size_t)(s0 + qq) * pitch, (nvar + 2) & ~1, lane);
#pragma unroll
      for (int qq = 0; qq < PF0; qq++) {
        const int s = s0 + qq;
        if (s >= ni || wide) break;
        const RowRegs<T, 1> &r = rr[qq];
        const bool fits = ((uabs64(r.v[0][0]) | uabs64(r.v[0][1])) >> 31) == 0;  // below 2^31 in magnitude
        if (ballot64(!fits)) {
          wide = true;
          break;
        }
        RowRegs32<1> z;
        z.v[0][0] = (int)r.v[0][0];
        z.v[0][1] = (int)r.v[0][1];
        row_store32p(z, vals + (size_t)s * W, lane, W);
        npacked = s + 1;
        const bool den1 = S.den[s] == 1;
        const int red = g_sig ? (g_sig[s] & SIG_RED) : (den1 ? SIG_RED : 0);
        mcw = max(mcw, lean_publish(z, S, cst, s, -1, red, lane, nvar));
        if (tflags & PIPAMD_T_SORT) {
          // traiter.c:576-589: size = max_j |(int)(v_j / den)| over the unknowns (as pip_advance_kernel computes it)
          int sz = 0;
          if (den1) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int q2 = z.v[0][h];
              const int aq = q2 < 0 ? (int)(0u - (unsigned)q2) : q2;
              if (2 * lane + h < nvar) sz = sz > aq ? sz : aq;
            }
          } else {
            const double d = to_double(S.den[s]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int q2 = trunc_int_x86((double)z.v[0][h] / d);
              const int aq = q2 < 0 ? (int)(0u - (unsigned)q2) : q2;
              if (2 * lane + h < nvar) sz = sz > aq ? sz : aq;
            }
          }
          const unsigned szw = wave_minmax_u32<true>((unsigned)sz);
          if (lane == 0) {
            S.size[s] = (float)szw;
            if ((int)S.srow[s] >= nvar) atomicMax(&sc.smaxbits, (u64)szw);
          }
        }
      }
    }
    if (wide) {
      // an entry beyond 32 bits: not a job for this kernel.  Its header is untouched (FRESHROWS and SORT still stand);
      // rows that came from the block itself and were already rewritten as ints are widened again.
      if (!fresh) rows_unpack(vals, npacked, lane, W, S.rcls);
      if (lane == 0 && q.out_count) {
        q.out_list[atomicAdd(q.out_count, 1)] = jb;
        atomicMax(q.out_maxni, ni);
      }
      return;
    }
  }
  tflags &= ~PIPAMD_T_FRESHROWS;
  __builtin_amdgcn_wave_barrier();
  if (tflags & PIPAMD_T_SORT) {
    sort_rows(S, nvar, nligne, (double)sc.smaxbits);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nligne; i += 64)
      if (!(S.ref[i] & UNITBIT)) S.srow[S.ref[i]] = (u16)i;
    tflags &= ~PIPAMD_T_SORT;
    // the sort keys overwrote the constant terms: back from the rows (column 127)
    __threadfence_block();
    for (int s = lane; s < ni; s += 64) cst[s] = reinterpret_cast<const int *>(vals + (size_t)s * W)[nvar];
    __builtin_amdgcn_wave_barrier();
  }
  for (int s = lane; s < ni; s += 64) {
    const int ff = S.fl[s];
    if (ff & PIPAMD_F_MINUS)
      atomicMin(&sc.pivi, (int)S.srow[s]);
    else if (ff == PIPAMD_F_UNKNOWN) {
      const int ec = exam_class(S.sig[s]);
      S.nf[s] = (u8)ec;
      if (ec == PIPAMD_F_MINUS) atomicMin(&sc.pivi2, (int)S.srow[s]);
    }
  }
  __builtin_amdgcn_wave_barrier();

  PROF(10);
  int status = PIPAMD_ST_RUN;
  int why = 0;  // why a job left this kernel unfinished (PipJob.pad_, read by tools/lean_split.py): 1 pivot budget,
                // 2 a row beyond ints, 3 a cut's denominator, 5 no room in the LDS image
  for (int iter = 0;; iter++) {
    why = 1;
    if (iter >= iter_limit) break;  // status stays RUN: the next launch resumes the job
    if (nlog >= LOGCAP) break;
    why = 2;
    if (mcw > 1) break;  // a row left the ints (it is stored in the general format): the general kernel goes on
    why = 0;
    int pivi = sc.pivi;
    if (pivi == BIG_I) {
      // -------------- exam_coef (its flags were prepared by phase C), then integrer if nothing is negative
      pivi = sc.pivi2;
      for (int s = lane; s < ni; s += 64)
        if (S.fl[s] == PIPAMD_F_UNKNOWN && (int)S.srow[s] <= pivi) S.fl[s] = S.nf[s];
      __builtin_amdgcn_wave_barrier();
      if (pivi == BIG_I) {
        if (!(tflags & PIPAMD_T_INT)) {
          status = PIPAMD_ST_SOLUTION;
          break;
        }
        // ------------- integrer(): first non-integral row among the unknowns (integrer.c:305-486, constant cuts)
        if (lane == 0) sc.tmp = BIG_I;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nvar; i += 64) {
          const int rf = S.ref[i];
          if (rf & UNITBIT) continue;
          const T D = S.den[rf];
          if (D == 1) continue;
          if (wneg(fmod64(wneg((T)cst[rf]), D)) != 0) atomicMin(&sc.tmp, i);
        }
        __builtin_amdgcn_wave_barrier();
        const int ci = sc.tmp;
        if (ci == BIG_I) {
          status = PIPAMD_ST_SOLUTION;
          break;
        }
        const int cslot = S.ref[ci];
        const T D64 = uni64(S.den[cslot]);
        why = 3;
        if (D64 <= 0 || D64 >= ((T)1 << 31)) break;  // the cut's entries (below D) might not be ints: the general kernel goes on
        const int D = (int)D64;
        RowRegs32<1> r;
        row_load32p(r, vals + (size_t)cslot * W, lane, W);
        bool okv = false;
        const bool tinyD = D < (1 << 15) && S.rcls[cslot] == 0;  // |v| < 2^15 and D < 2^15: remainders through a float reciprocal
        const float rD = __builtin_amdgcn_rcpf((float)D);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int j = 2 * lane + h;
          const int v = r.v[0][h];
          // piplib_llmod (integrer.c:69-74): the remainder in [0, D)
          int pos;
          if (tinyD) {
            const unsigned m = umod_tiny((unsigned)(v < 0 ? -v : v), (unsigned)D, rD);
            pos = v < 0 ? (m ? D - (int)m : 0) : (int)m;  // v mod D
          } else {
            const int m = v % D;
            pos = m < 0 ? m + D : m;
          }
          int x;
          if (j < nvar) {
            x = pos;
            okv |= x > 0;
          } else {
            x = pos ? pos - D : 0;  // -((-v) mod D) == (v mod D) - D unless D divides v
          }
          r.v[0][h] = x;
        }
        const bool any_v = ballot64(okv) != 0;
        int verdict;
        if (!any_v)
          verdict = PIPAMD_ST_NIL;  // integrer.c:482-485 case (b)
        else if (ni >= cap_ni)
          verdict = PIPAMD_ST_CAPACITY;
        else if (ni >= Smax || nligne >= Lmax)
          verdict = -1;  // no room in this launch's LDS image: pause
        else {
          verdict = PIPAMD_ST_RUN;
          row_store32p(r, vals + (size_t)ni * W, lane, W);
          mcw = max(mcw, lean_publish(r, S, cst, ni, -1, 0, lane, nvar));
          if (lane == 0) {
            S.fl[ni] = PIPAMD_F_MINUS;
            S.nf[ni] = 0;
            S.den[ni] = D64;
            S.ref[nligne] = (u16)ni;
            S.srow[ni] = (u16)nligne;
          }
        }
        if (lane == 0) sc.aux = ci;
        __builtin_amdgcn_wave_barrier();
        why = 5;
        if (verdict != PIPAMD_ST_RUN) {
          status = verdict < 0 ? PIPAMD_ST_RUN : verdict;
          break;
        }
        pivi = nligne;
        ni++;
        nligne++;
      }
    }
    PROF(0);
    // ---------------- A: pivot row, choisir_piv, work list
    const int pslot = S.ref[pivi];
    const T dpiv = uni64(S.den[pslot]);
    // small path for a row: the row and the pivot row in class 0 and the pivot row's denominator below 2^15 (then the
    // multipliers are below 2^15 as well and every product below 2^30)
    const bool psmall = S.rcls[pslot] == 0 && dpiv > -((T)1 << 15) && dpiv < ((T)1 << 15);
    npiv++;
    RowRegs32<1> pr;
    row_load32p(pr, vals + (size_t)pslot * W, lane, W);
    const int psig_v = S.sig[pslot];
#ifdef PIP_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PROF(1);
    const int pj = mcw == 0 ? choose_column32<true>(S, pr, vals, W, nvar, nligne, pivi, &sc)
                            : choose_column32<false>(S, pr, vals, W, nvar, nligne, pivi, &sc);
    if (pj == -1) {  // traiter.c:782-785
      status = PIPAMD_ST_NIL;
      break;
    }
    PROF(2);
    const int pe = pj & 1, pl = pj >> 1;
    int nwork = 0;
    for (int s0 = 0; s0 < ni; s0 += 64) {
      const int s = s0 + lane;
      bool need = false;
      if (s < ni) {
        if (s == pslot)
          need = true;
        else {
          const bool nzb = (S.nzm[(size_t)s * NM + pe] >> pl) & 1;
          if (nzb || !(S.sig[s] & SIG_RED))
            need = true;
          else
            S.sig[s] &= ~0xC0;  // entry in the pivot column is 0: sign hint "zero"
        }
      }
      const u64 m = ballot64(need);
      if (need) S.work[nwork + __popcll(m & ((1ull << lane) - 1))] = (u16)s;
      nwork += __popcll(m);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {  // phase C refills them
      sc.pivi = BIG_I;
      sc.pivi2 = BIG_I;
    }
    const int pivj = pj;
    const int pivot = __builtin_amdgcn_readlane(pr.v[0][0], pl) * (1 - pe) + __builtin_amdgcn_readlane(pr.v[0][1], pl) * pe;
    if (lane == 0) {
      g_log[2 * nlog] = (T)pivot;
      g_log[2 * nlog + 1] = dpiv;
    }
    nlog++;
    const int ku = S.urow[pivj];  // unit row of the entering column
    const int pred = psig_v & SIG_RED;
    PROF(3);
    // ---------------- B: eliminate the pivot column
    nupd += nwork - 1;
    {
      // a queue of PF rows on their way from HBM / L2 (a row is two registers here): the row at its head is updated
      // while the loads behind it are in flight
      constexpr int PF = PIP_LEAN_PF;
      RowRegs32<1> rq[PF];
      int sq[PF];
#pragma unroll
      for (int q2 = 0; q2 < PF; q2++) {
        sq[q2] = S.work[q2 < nwork ? q2 : 0];
        if (q2 < nwork && sq[q2] != pslot) row_load32p(rq[q2], vals + (size_t)sq[q2] * W, lane, W);
      }
      // while the first rows are on their way: the pivot slot is recycled for the row replacing ku's unit row
      // (traiter.c:461-465,503-513) -- it needs no load, the pivot row is in registers
      if (dpiv > -((T)1 << 31) && dpiv < ((T)1 << 31)) {
        RowRegs32<1> r;
#pragma unroll
        for (int h = 0; h < 2; h++) r.v[0][h] = (2 * lane + h == pivj) ? (int)dpiv : -pr.v[0][h];
        row_store32p(r, vals + (size_t)pslot * W, lane, W);
        mcw = max(mcw, lean_publish(r, S, cst, pslot, pivj, pred, lane, nvar));
      } else {  // the denominator is no int: that row is not one either
        i64 zw[2];
#pragma unroll
        for (int h = 0; h < 2; h++) zw[h] = (2 * lane + h == pivj) ? dpiv : -(i64)pr.v[0][h];
        row_store64w(zw, vals + (size_t)pslot * W, lane, W);
        mcw = max(mcw, lean_publish_wide(zw, S, cst, pslot, pivj, pred, lane, nvar));
      }
      PROF(4);
      for (int w = 0; w < nwork; w++) {
        const int s = sq[0];
        RowRegs32<1> r = rq[0];
#pragma unroll
        for (int q2 = 0; q2 + 1 < PF; q2++) {
          rq[q2] = rq[q2 + 1];
          sq[q2] = sq[q2 + 1];
        }
        if (w + PF < nwork) {
          sq[PF - 1] = S.work[w + PF];
          if (sq[PF - 1] != pslot) row_load32p(rq[PF - 1], vals + (size_t)sq[PF - 1] * W, lane, W);
        }
        T *row = vals + (size_t)s * W;
        {
          if (s == pslot) continue;
#ifdef PIP_PROFILE
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF - 1) : "memory");
#endif
          PROF(5);
          // multipliers from the row's own pivot-column entry (traiter.c:470-476); ints
          int foo = __builtin_amdgcn_readlane(r.v[0][0], pl) * (1 - pe) + __builtin_amdgcn_readlane(r.v[0][1], pl) * pe;
          const T den_s = uni64(S.den[s]);
          int lp = pivot;
          T g0 = den_s;
          if (pivot != 1) {
            const unsigned d = gcd_u32((unsigned)pivot, (unsigned)(foo < 0 ? -foo : foo));
            if (d != 1) {  // (d == 0 cannot be: pivot > 0)
              lp = (int)exact_quo<i64>((i64)pivot, (i64)d);
              foo = (int)exact_quo<i64>((i64)foo, (i64)d);
            }
            g0 = wmul((T)lp, den_s);
          }
          T nd;
          PROF(6);
          if (psmall && S.rcls[s] == 0) {
            // small path: every operand below 2^15, every product below 2^30
            int z[1][2];
            unsigned mx = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
              int v = __mul24(r.v[0][h], lp) - __mul24(pr.v[0][h], foo);
              if (2 * lane + h == pivj) v = __mul24((int)dpiv, foo);
              z[0][h] = v;
              mx |= (unsigned)(v < 0 ? -v : v);
            }
            if (!small_reduce<1>(z, mx, g0, lane, nd)) {
              if (lane == 0) sc.bad = 1;
            }
            r.v[0][0] = z[0][0];
            r.v[0][1] = z[0][1];
            PROF(7);
            row_store32p(r, row, lane, W);
            mcw = max(mcw, lean_publish(r, S, cst, s, pivj, SIG_RED, lane, nvar));
          } else {
            // mid path: int operands, products below 2^62 in long longs -- pip_advance_kernel's update_row on the same
            // values (its wrap-around arithmetic has nothing to wrap here, except dpiv * foo under a denominator
            // beyond ints, which wraps the same way)
            i64 zw[2];
            u64 mx = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
              i64 v = (i64)r.v[0][h] * (i64)lp - (i64)pr.v[0][h] * (i64)foo;
              if (2 * lane + h == pivj) v = wmul(dpiv, (i64)foo);
              zw[h] = v;
              mx |= uabs64(v);
            }
#if PIP_LEAN_MID_INV
            if (!row_reduce<i64, 2, false>(zw, mx, g0, lane, nd, wmul(dpiv, (i64)foo))) {
#else
            if (!row_reduce_rem<i64, 2>(zw, mx, g0, lane, nd)) {  // (the remainder loop: reduce_by_inverse costs this kernel 12 bytes of scratch)
#endif
              if (lane == 0) sc.bad = 1;
            }
            if (ballot64(((uabs64(zw[0]) | uabs64(zw[1])) >> 31) != 0) == 0) {
              r.v[0][0] = (int)zw[0];
              r.v[0][1] = (int)zw[1];
              row_store32p(r, row, lane, W);
              mcw = max(mcw, lean_publish(r, S, cst, s, pivj, SIG_RED, lane, nvar));
            } else {  // not an int row any more: general format, the lean run ends after this pivot
              row_store64w(zw, row, lane, W);
              mcw = max(mcw, lean_publish_wide(zw, S, cst, s, pivj, SIG_RED, lane, nvar));
            }
          }
          if (lane == 0) S.den[s] = nd;
          PROF(8);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    PROF(4);
    if (sc.bad) {
      status = PIPAMD_ST_OVERFLOW;
      break;
    }
    // ---------------- C: swap roles, refresh the sign hints, next chercher (traiter.c:503-529)
    if (lane == 0) {
      S.ref[pivi] = (u16)(UNITBIT | UNITZERO | pivj);
      S.urow[pivj] = (u16)pivi;
    }
    for (int s = lane; s < ni; s += 64) {
      int ff, k;
      if (s == pslot) {
        k = ku;
        ff = PIPAMD_F_PLUS;
        S.den[s] = (T)pivot;
        S.srow[s] = (u16)ku;
        S.ref[ku] = (u16)s;
      } else {
        k = S.srow[s];
        ff = S.fl[s];
      }
      const int sg = S.sig[s];
      const int ps = SIG_PIV(sg);
      const int fff = ps == 1 ? PIPAMD_F_PLUS : (ps == 2 ? PIPAMD_F_MINUS : PIPAMD_F_ZERO);
      if (fff != PIPAMD_F_ZERO && fff != ff) {
        if (ff == PIPAMD_F_ZERO)
          ff = (fff == PIPAMD_F_MINUS) ? PIPAMD_F_UNKNOWN : fff;
        else
          ff = PIPAMD_F_UNKNOWN;
      }
      S.fl[s] = (u8)ff;
      if (ff & PIPAMD_F_MINUS)
        atomicMin(&sc.pivi, k);
      else if (ff == PIPAMD_F_UNKNOWN) {
        const int ec = exam_class(sg);
        S.nf[s] = (u8)ec;
        if (ec == PIPAMD_F_MINUS) atomicMin(&sc.pivi2, k);
      }
    }
    __builtin_amdgcn_wave_barrier();
    PROF(9);
  }

  // ---- epilogue: the row tables, the header and (if any) the solution, as pip_advance_kernel writes them
  __builtin_amdgcn_wave_barrier();
  {
    const int L = J->L;
    T *g_den = (T *)(arena + J->rows_off);
    int *g_flag = (int *)(g_den + L);
    int *g_ref = g_flag + L;
    for (int i = lane; i < nligne; i += 64) {
      const int rf = S.ref[i];
      if (rf & UNITBIT) {
        g_den[i] = 1;
        g_flag[i] = PIPAMD_F_UNIT | ((rf & UNITZERO) ? PIPAMD_F_ZERO : 0);
        g_ref[i] = UNITCOL(rf);
      } else {
        g_den[i] = S.den[rf];
        g_flag[i] = S.fl[rf];
        g_ref[i] = rf;
      }
    }
  }
  tflags &= ~PIPAMD_T_STATE;
  if (status == PIPAMD_ST_RUN) {
    const int Sl = J->S;
    u64 *g_nzm = (u64 *)(arena + J->state_off);
    u16 *g_sig = (u16 *)(g_nzm + (size_t)Sl * NM);
    u8 *g_rcls = (u8 *)(g_sig + Sl);
    for (int s = lane; s < ni; s += 64) {
      g_sig[s] = S.sig[s];
      g_rcls[s] = S.rcls[s];
    }
    for (int e = lane; e < ni * NM; e += 64) g_nzm[e] = S.nzm[e];
    tflags |= PIPAMD_T_STATE;
  }
  if (status == PIPAMD_ST_SOLUTION) {
    // solution(), traiter.c:255-271: the constant column of rows 0..nvar-1
    T *sol_num = (T *)(arena + J->sol_off);
    T *sol_den = sol_num + nvar;
    for (int i = lane; i < nvar; i += 64) {
      const int rf = S.ref[i];
      T v = 0, d = 1;
      if (!(rf & UNITBIT)) {
        v = (T)cst[rf];  // (the constant terms are kept current in LDS by lean_publish)
        d = S.den[rf];
      }
      sol_num[i] = v;
      sol_den[i] = d;
    }
  }
  if (status == PIPAMD_ST_RUN || status == PIPAMD_ST_CAPACITY) {
    // the job goes on elsewhere (pip_advance_kernel, pip_rehouse_kernel): its rows in the general format again
    rows_unpack(vals, ni, lane, W, S.rcls);
  }
  int mc = 0;
  for (int s = lane; s < ni; s += 64)
    if (S.rcls[s] > mc) mc = S.rcls[s];
  mc = ballot64(mc == 3) ? 3 : (ballot64(mc == 2) ? 2 : (ballot64(mc == 1) ? 1 : 0));
  if (lane == 0) {
    J->ni = ni;
    J->npiv = npiv;
    J->ncut = ncut0 + ni;
    J->nupd = nupd;
    J->nlog = nlog;
    J->pad_ = why;
    J->tflags = tflags;
    J->state_nch = 1;
    J->maxabs = (u64)mc;
    J->aux = sc.aux;
    J->status = status;
    if (status == PIPAMD_ST_RUN && q.out_count) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, ni);
    }
    if (status == PIPAMD_ST_CAPACITY && q.out_count) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, PIPAMD_Q_CAPFLAG | ni);
      atomicAdd(q.out_maxni + 1, 1);  // (the list\'s third control word: tableaux out of rows)
    }
  }
  PROF(11);
#ifdef PIP_PROFILE
  PROF_FLUSH(prof);
#endif
}

template <int SC, bool FULL>
hipError_t launch_lean(const AdvanceLaunch &a) {
  const int grid = a.grid > 0 && a.grid < a.njobs ? a.grid : a.njobs;
  const size_t shm = lean_lds_bytes(SC);
#ifdef PIP_PROFILE
  hipLaunchKernelGGL((pip_lean_kernel<SC, FULL>), dim3(grid), dim3(64), shm, a.stream, a.jobs, a.arena, a.njobs, a.iter_limit, a.q,
                     (u64 *)a.prof);
#else
  hipLaunchKernelGGL((pip_lean_kernel<SC, FULL>), dim3(grid), dim3(64), shm, a.stream, a.jobs, a.arena, a.njobs, a.iter_limit, a.q);
#endif
  return hipGetLastError();
}
// the row-capacity classes of launch_static (pip_kernels.hip)
#define PIP_LEAN_CLASSES(X) \
  X(64, true) X(96, true) X(112, true) X(128, true) X(160, true) X(64, false) X(96, false) X(112, false) X(128, false) X(160, false)
#define PIP_LEAN_DEFINE(SC, FULL) template hipError_t launch_lean<SC, FULL>(const AdvanceLaunch &);
#endif  // PIP_LEAN_H
