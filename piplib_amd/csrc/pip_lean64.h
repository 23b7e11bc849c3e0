// piplib_amd/csrc/pip_lean64.h -- the lean bulk kernel of the 128-bit Entier flavour: pip_lean_kernel's design (pip_lean.h)
// one width up.
//
// The overflow-safe flavour (piplib.h:42-88: the whole library built with a wider Entier) exists for the tableaux on
// which 64-bit arithmetic overflows -- but what outgrows 64 bits there are the determinant limbs and the products of a
// row update, seldom the rows themselves: of the 1,000 tableaux of BASELINE's configs[4] (the batch pinned by
// tests/golden/gmp/wide128.json) the reference's GMP build forms a value beyond 2^63 on 587, yet on two in three every
// STORED entry stays below 2^63 from the first pivot to the last (measured with this kernel: 657 of the 1,000 finish in
// it, 325 leave on a row beyond 2^63).  pip_advance_kernel<__int128> keeps every row as 128-bit entries: 16 registers a
// row of 256 columns, 128 VGPRs, 400 bytes of scratch per lane, four waves a tableau of which three wait while one runs
// choisir_piv.  This kernel is the same loop for the regime those tableaux live in and nothing else (OPT-IN,
// pipamd_engine_set_lean64: measured on that batch it is no faster than the four-wave 128-bit kernel): one wave per tableau, no parameters, 129 ... 256 columns, plain cuts, rows skipped -- and
// EVERY entry of EVERY row below 2^63 in magnitude, i.e. a long long.  Under that invariant
//   * rows live in HBM as long longs (8 W bytes, the first half of the row's slot of W 128-bit entries), lane l holds
//     columns l, 64 + l, 128 + l, 192 + l (the geometry of pip_advance_kernel<__int128, 4>, so that the saved summaries
//     of a paused job mean the same to both kernels): half the traffic, half the registers;
//   * while the rows involved are in magnitude class 0 (entries below 2^31; pivot row's denominator too) every product
//     of a pivot fits 63 bits: the elimination is 64-bit arithmetic ("small" path); else the products are 64 x 64 -> 128
//     bits (four 32-bit multiply-adds each, not ten) and the row gcd / division run on 128-bit values ("mid" path) --
//     reduce_by_inverse picks the narrowest width that holds them.  The result is a row of long longs again, almost always;
//   * choisir_piv's cross products are 64-bit while every row is in class 0, 128-bit else.
// A rewritten row that does NOT fit long longs any more is stored in the general format (the whole slot) and the lean run
// ends after that pivot; a tableau that leaves -- or anything else this kernel does not do -- is handed over in the
// general format (rows widened in place, the row tables and saved summaries of a paused job of
// pip_advance_kernel<__int128, 4>) and stays PIPAMD_ST_RUN on the launch list.  Same algorithm, same statuses, same bits
// as pip_advance_kernel -- the reference's traiter()/pivoter()/choisir_piv()/exam_coef()/integrer()/tab_sort_rows
// (traiter.c:101-159, 297-548, 556-623, 628-791; integrer.c:305-486) -- checked tableau by tableau against the 128-bit
// oracle and the reference's GMP build (tests/test_gpu_parity.py: test_lean64_kernel_paths, test_full_size_int128_config).
#ifndef PIP_LEAN64_H
#define PIP_LEAN64_H
#include "pip_advance.h"

#ifndef PIP_LEAN64_WAVES
#define PIP_LEAN64_WAVES 3  // waves per SIMD the kernel is bounded to (168 VGPRs)
#endif

struct Row64 {
  i64 v[4];  // lane l: columns l, 64 + l, 128 + l, 192 + l
};

__device__ __forceinline__ void row_load64p(Row64 &r, const i128 *slot, int lane, int W) {
  const i64 *p = reinterpret_cast<const i64 *>(slot);
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int j = 64 * c + lane;
    r.v[c] = j < W ? p[j] : 0;
  }
}
__device__ __forceinline__ void row_store64p(const Row64 &r, i128 *slot, int lane, int W) {
  i64 *p = reinterpret_cast<i64 *>(slot);
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int j = 64 * c + lane;
    if (j < W) p[j] = r.v[c];
  }
}
// a row that no longer fits long longs: the general format, the whole slot
__device__ __forceinline__ void row_store128w(const i128 (&z)[4], i128 *slot, int lane, int W) {
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int j = 64 * c + lane;
    if (j < W) {
      longlong2 t;
      t.x = (i64)(u64)(u128)z[c];
      t.y = (i64)(u64)((u128)z[c] >> 64);
      *reinterpret_cast<longlong2 *>(slot + j) = t;
    }
  }
}
// rows [0, n) of a block, packed -> the general format, each within its own slot (a row's loads are back before its slot
// is overwritten); rows of class 2 or 3 (rcls, LDS) are in the general format already
__device__ __forceinline__ void rows_unpack64(i128 *vals, int n, int lane, int W, const u8 *rcls) {
  for (int s0 = 0; s0 < n; s0 += 2) {
    Row64 rr[2];
    bool packed[2];
#pragma unroll
    for (int qq = 0; qq < 2; qq++) {
      packed[qq] = s0 + qq < n && rcls[s0 + qq] < 2;
      if (packed[qq]) row_load64p(rr[qq], vals + (size_t)(s0 + qq) * W, lane, W);
    }
#pragma unroll
    for (int qq = 0; qq < 2; qq++)
      if (packed[qq]) {
        i128 z[4];
#pragma unroll
        for (int c = 0; c < 4; c++) z[c] = (i128)rr[qq].v[c];
        row_store128w(z, vals + (size_t)(s0 + qq) * W, lane, W);
      }
  }
}

// sign summary, non-zero bitmap and magnitude class (0: every entry below 2^31, 1: below 2^63) of a packed row of nvar
// unknowns + constant; lane 0 publishes them for slot s.  Returns the class.
__device__ __forceinline__ int lean64_publish(const Row64 &z, const Shared<i128> &S, i64 *cst, int s, int pivj, int extra_sig, int lane,
                                              int nvar) {
  i64 pick = 0;
#pragma unroll
  for (int c = 0; c < 4; c++)
    if (c == (nvar >> 6)) pick = z.v[c];
  const i64 cz = readlane64(pick, nvar & 63);
  int sig = extra_sig | sign_code(cz);
  if (pivj >= 0) {
    i64 pp = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (c == (pivj >> 6)) pp = z.v[c];
    sig |= sign_code(readlane64(pp, pivj & 63)) << 6;
  }
  u64 mx = 0, nz[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    mx |= uabs64(z.v[c]);
    nz[c] = ballot64(z.v[c] != 0);
  }
  const int cls = ballot64((mx >> 31) != 0) ? 1 : 0;
  if (lane == 0) {
    S.sig[s] = (u16)sig;
    S.rcls[s] = (u8)cls;
    cst[s] = cz;
#pragma unroll
    for (int c = 0; c < 4; c++) S.nzm[(size_t)s * 4 + c] = nz[c];
  }
  return cls;
}
// the same for a row that left the long longs: pip_advance_kernel<__int128>'s classes (2: below 2^95, 3: beyond); the
// constant term kept here is truncated -- the lean run ends before anything reads it
__device__ __forceinline__ int lean64_publish_wide(const i128 (&z)[4], const Shared<i128> &S, i64 *cst, int s, int pivj, int extra_sig,
                                                   int lane, int nvar) {
  i128 pick = 0;
#pragma unroll
  for (int c = 0; c < 4; c++)
    if (c == (nvar >> 6)) pick = z[c];
  const i128 cz = readlane64(pick, nvar & 63);
  int sig = extra_sig | sign_code(cz);
  if (pivj >= 0) {
    i128 pp = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (c == (pivj >> 6)) pp = z[c];
    sig |= sign_code(readlane64(pp, pivj & 63)) << 6;
  }
  u128 mx = 0;
  u64 nz[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    mx |= uabs64(z[c]);
    nz[c] = ballot64(z[c] != 0);
  }
  int cls = cls_of<i128>(mx);
  if (cls < 2) cls = 2;  // (it does not fit a long long: at least 2^63)
  if (lane == 0) {
    S.sig[s] = (u16)sig;
    S.rcls[s] = (u8)cls;
    cst[s] = (i64)cz;
#pragma unroll
    for (int c = 0; c < 4; c++) S.nzm[(size_t)s * 4 + c] = nz[c];
  }
  return cls;
}

// bytes of this kernel's LDS image for S row slots and L logical rows
__host__ __device__ constexpr size_t lean64_lds_bytes(int S, int L) {
  return ((size_t)(16 + 8 + 32 + 2 + 2 + 2 + 3) * S + 2 * (size_t)L + 2 * 256 + 15) & ~(size_t)15;
}

// choisir_piv (traiter.c:297-341) as choose_column<__int128, 4> does it, on packed rows.  SMALL: every row of the tableau
// is in class 0 (entries below 2^31), the cross products fit long longs; else they are below 2^126 and their difference
// a 128-bit number.
template <bool SMALL>
__device__ __forceinline__ int choose_column64(const Shared<i128> &S, const Row64 &prow, const i128 *vals, int W, int nvar, int nligne,
                                               int pivi, Scalars *sc) {
  constexpr int NM = 4;
  const int lane = threadIdx.x & 63;
  i64 a[4];
  int u[4];
  bool cand[4];
  u64 cm[NM];
  int count = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int j = 64 * c + lane;
    a[c] = j < nvar ? prow.v[c] : 0;
    cand[c] = a[c] > 0;
    u[c] = cand[c] ? (int)S.urow[j] : -1;
    cm[c] = ballot64(cand[c]);
    count += __popcll(cm[c]);
  }
  if (count == 0) return -1;
  for (int k0 = 0; k0 < nligne && count > 1; k0 += 64) {
    const int k = k0 + lane;
    bool rel = false;
    if (k < nligne && k != pivi) {
      const int rf = S.ref[k];
      if (!(rf & UNITBIT)) {
        const u64 *m = S.nzm + (size_t)rf * NM;
        rel = ((m[0] & cm[0]) | (m[1] & cm[1]) | (m[2] & cm[2]) | (m[3] & cm[3])) != 0;
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
      for (int c = 0; c < 4; c++) nel += __popcll(ballot64(cand[c] && u[c] < kk));
      if (nel == count) goto last_unit_wins;
      if (nel) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if (u[c] < kk) cand[c] = false;
          cm[c] = ballot64(cand[c]);
        }
        count -= nel;
        if (count == 1) break;
      }
      {
        const u64 *m = S.nzm + (size_t)sl * NM;
        if (!((m[0] & cm[0]) | (m[1] & cm[1]) | (m[2] & cm[2]) | (m[3] & cm[3]))) continue;  // cannot separate them
      }
      // real row kk: keep the minimal ratios
      Row64 n;
      row_load64p(n, vals + (size_t)sl * W, lane, W);
      for (;;) {
        // reference column b = first remaining candidate
        i64 ab = 0, nb = 0;
        {
          int cb = 3;
#pragma unroll
          for (int c = 3; c >= 0; c--)
            if (cm[c]) cb = c;
          const int src = __ffsll((long long)cm[cb]) - 1;
          i64 pa = 0, pn = 0;
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (c == cb) {
              pa = a[c];
              pn = n.v[c];
            }
          ab = readlane64(pa, src);
          nb = readlane64(pn, src);
        }
        bool neg[4];
        int nneg = 0, nzero = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          bool xneg, xzero;
          if constexpr (SMALL) {
            const i64 x = ab * n.v[c] - nb * a[c];
            xneg = x < 0;
            xzero = x == 0;
          } else {
            const i128 x = (i128)ab * (i128)n.v[c] - (i128)nb * (i128)a[c];
            xneg = x < 0;
            xzero = x == 0;
          }
          neg[c] = cand[c] && xneg;
          const bool zero = cand[c] && xzero;
          nneg += __popcll(ballot64(neg[c]));
          nzero += __popcll(ballot64(zero));
          if (!neg[c] && !zero) cand[c] = false;  // strictly larger: out
        }
        if (nneg == 0) {
          count = nzero;
        } else {
#pragma unroll
          for (int c = 0; c < 4; c++) cand[c] = neg[c];
          count = nneg;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) cm[c] = ballot64(cand[c]);
        if (nneg == 0 || count == 1) break;
      }
    }
  }
  if (count == 1) {
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (cm[c]) return 64 * c + __ffsll((long long)cm[c]) - 1;
  }
last_unit_wins:
  // only unit rows left to look at: the column whose unit row comes last survives
  if (lane == 0) sc->tmp2 = -1;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < 4; c++)
    if (cand[c]) atomicMax(&sc->tmp2, (u[c] << 10) | (64 * c + lane));
  __builtin_amdgcn_wave_barrier();
  return sc->tmp2 & 1023;
}

__global__ __launch_bounds__(64, PIP_LEAN64_WAVES) void pip_lean64_kernel(PipJob *jobs, i64 *arena, int njobs, int Smax, int Lmax,
                                                                          int iter_limit, PipQueue q
#ifdef PIP_PROFILE
                                                                          , u64 *prof
#endif
) {
  typedef i128 T;
  // (diagnostic build only, tools/dbg_prof_lean64.py: cycle stamps per piece of the loop -- 0 exam/integrer, 1 pivot row
  // load, 2 choisir_piv, 3 work list, 4 recycled slot + barrier, 5 wait for a work row, 6 multipliers, 7 products + row gcd
  // + division, 8 store + summary, 9 phase C, 10 entry, 11 epilogue)
  PROF_DECL;
  constexpr int WP = 256, NM = 4;
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
      atomicAdd(q.out_maxni + 1, 1);
    }
    return;
  }
  int tflags = J->tflags;
  int ni = J->ni;
  const int nvar = J->nvar, W = J->W;
  int nligne = nvar + ni;
  // what this kernel does not do stays with pip_advance_kernel: the job goes on the launch list untouched
  const bool mine = nvar < 256 && nvar >= 1 && J->nparm == 0 && J->bigparm < 0 && W > 128 && W <= 256 && J->ebits == 128 &&
                    !(tflags & (PIPAMD_T_NOSKIP | PIPAMD_T_DEEPEST)) && ni <= Smax && nligne <= Lmax &&
                    (!(tflags & PIPAMD_T_STATE) || J->state_nch == 4);
  if (!mine) {
    if (lane == 0 && q.out_count) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, ni);
    }
    return;
  }
  T *vals = (T *)(arena + J->vals_off);
  const int ncut0 = J->ncut - ni;  // cuts so far = ncut0 + ni (every row this kernel appends is a cut)
  const int cap_ni = min(J->S, J->L - nvar);  // rows the job's block holds
  int npiv = J->npiv, nupd = J->nupd;
  T *g_log = (T *)(arena + J->log_off);
  constexpr int LOGCAP = PIPAMD_DETLOG;
  int nlog = J->nlog;

  Shared<T> S;  // the tables of pip_advance_kernel's image this kernel uses
  i64 *cst;     // [S] constant terms (long longs here); the entry-time sort keys share their storage
  {
    unsigned char *p = smem;
    S.den = (T *)p;      p += sizeof(T) * Smax;
    S.nzm = (u64 *)p;    p += sizeof(u64) * (size_t)Smax * NM;
    cst = (i64 *)p;
    S.size = (float *)p; p += sizeof(i64) * Smax;
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

  // ---- one pass over the tableau: the rows become long longs, summaries, sort keys.  A row with an entry of 2^63 or
  // more: not a job for this kernel.
  int mcw = 0;  // largest magnitude class published so far: 0 small path everywhere, 1 long long rows, beyond: the lean run ends
  {
    // a job that paused in an earlier launch: what the entry pass cannot see in the rows -- "gcd(row, denominator) is
    // known to be 1" -- comes from the saved summaries
    const u16 *g_sig = (tflags & PIPAMD_T_STATE) ? (const u16 *)((const u64 *)(arena + J->state_off) + (size_t)J->S * NM) : nullptr;
    int npacked = 0;
    bool wide = false;
    for (int s = 0; s < ni && !wide; s++) {
      RowRegs<T, 4> r;
      row_load<T, 4>(r, vals + (size_t)s * W, nvar + 1, lane);
      bool fits = true;
#pragma unroll
      for (int c = 0; c < 4; c++) fits &= (uabs64(r.v[c][0]) >> 63) == 0;
      if (ballot64(!fits)) {
        wide = true;
        break;
      }
      Row64 z;
#pragma unroll
      for (int c = 0; c < 4; c++) z.v[c] = (i64)r.v[c][0];
      row_store64p(z, vals + (size_t)s * W, lane, W);
      npacked = s + 1;
      const bool den1 = S.den[s] == 1;
      const int red = g_sig ? (g_sig[s] & SIG_RED) : (den1 ? SIG_RED : 0);
      mcw = max(mcw, lean64_publish(z, S, cst, s, -1, red, lane, nvar));
      if (tflags & PIPAMD_T_SORT) {
        // traiter.c:576-589: size = max_j |(int)(v_j / den)| over the unknowns (as pip_advance_kernel computes it)
        int sz = 0;
        if (den1) {
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const i64 v = z.v[c];
            const int q2 = (v == (i64)(int)v) ? (int)v : (int)0x80000000;
            const int aq = q2 < 0 ? (int)(0u - (unsigned)q2) : q2;
            if (64 * c + lane < nvar) sz = sz > aq ? sz : aq;
          }
        } else {
          const double d = to_double(S.den[s]);
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const int q2 = trunc_int_x86(to_double((T)z.v[c]) / d);
            const int aq = q2 < 0 ? (int)(0u - (unsigned)q2) : q2;
            if (64 * c + lane < nvar) sz = sz > aq ? sz : aq;
          }
        }
        const unsigned szw = wave_minmax_u32<true>((unsigned)sz);
        if (lane == 0) {
          S.size[s] = (float)szw;
          if ((int)S.srow[s] >= nvar) atomicMax(&sc.smaxbits, (u64)szw);
        }
      }
    }
    if (wide) {
      // an entry beyond 63 bits: not a job for this kernel.  Its header is untouched; the rows already rewritten as
      // long longs are widened again.
      rows_unpack64(vals, npacked, lane, W, S.rcls);
      if (lane == 0 && q.out_count) {
        q.out_list[atomicAdd(q.out_count, 1)] = jb;
        atomicMax(q.out_maxni, ni);
      }
      return;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (tflags & PIPAMD_T_SORT) {
    [[clang::always_inline]] sort_rows(S, nvar, nligne, (double)sc.smaxbits);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nligne; i += 64)
      if (!(S.ref[i] & UNITBIT)) S.srow[S.ref[i]] = (u16)i;
    tflags &= ~PIPAMD_T_SORT;
    // the sort keys overwrote the constant terms: back from the rows (column nvar)
    __threadfence_block();
    for (int s = lane; s < ni; s += 64) cst[s] = reinterpret_cast<const i64 *>(vals + (size_t)s * W)[nvar];
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
  int why = 0;  // why a job left this kernel unfinished (PipJob.pad_): 1 pivot budget, 2 a row beyond long longs, 3 a cut's
                // denominator, 5 no room in the LDS image
  for (int iter = 0;; iter++) {
    why = 1;
    if (iter >= iter_limit) break;  // status stays RUN: the next launch resumes the job
    if (nlog >= LOGCAP) break;
    why = 2;
    if (mcw > 1) break;  // a row left the long longs (it is stored in the general format): the general kernel goes on
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
        const T D128 = uni64(S.den[cslot]);
        why = 3;
        if (D128 <= 0 || D128 >= ((T)1 << 62)) break;  // the cut's entries (below D) might not be long longs: the general kernel goes on
        const i64 D = (i64)D128;
        Row64 r;
        row_load64p(r, vals + (size_t)cslot * W, lane, W);
        bool okv = false;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const int j = 64 * c + lane;
          // piplib_llmod (integrer.c:69-74): the remainder in [0, D)
          const i64 m = crem(r.v[c], D);
          const i64 pos = m < 0 ? m + D : m;
          i64 x;
          if (j < nvar) {
            x = pos;
            okv |= x > 0;
          } else {
            x = pos ? pos - D : 0;  // -((-v) mod D) == (v mod D) - D unless D divides v
          }
          r.v[c] = x;
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
          row_store64p(r, vals + (size_t)ni * W, lane, W);
          mcw = max(mcw, lean64_publish(r, S, cst, ni, -1, 0, lane, nvar));
          if (lane == 0) {
            S.fl[ni] = PIPAMD_F_MINUS;
            S.nf[ni] = 0;
            S.den[ni] = D128;
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
    const bool dpiv64 = fits64(dpiv);
    // small path for a row: the row and the pivot row in class 0 and the pivot row's denominator below 2^31 (then the
    // multipliers are below 2^31 as well and every product below 2^62)
    const bool psmall = S.rcls[pslot] == 0 && dpiv > -((T)1 << 31) && dpiv < ((T)1 << 31);
    npiv++;
    Row64 pr;
    row_load64p(pr, vals + (size_t)pslot * W, lane, W);
    const int psig_v = S.sig[pslot];
#ifdef PIP_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PROF(1);
    const int pj = mcw == 0 ? choose_column64<true>(S, pr, vals, W, nvar, nligne, pivi, &sc)
                            : choose_column64<false>(S, pr, vals, W, nvar, nligne, pivi, &sc);
    if (pj == -1) {  // traiter.c:782-785
      status = PIPAMD_ST_NIL;
      break;
    }
    PROF(2);
    const int pc = pj >> 6, pl = pj & 63;
    int nwork = 0;
    for (int s0 = 0; s0 < ni; s0 += 64) {
      const int s = s0 + lane;
      bool need = false;
      if (s < ni) {
        if (s == pslot)
          need = true;
        else {
          const bool nzb = (S.nzm[(size_t)s * NM + pc] >> pl) & 1;
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
    i64 pivot;
    {
      i64 pp = 0;
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (c == pc) pp = pr.v[c];
      pivot = readlane64(pp, pl);
    }
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
      // the pivot slot is recycled for the row replacing ku's unit row (traiter.c:461-465,503-513) -- it needs no load,
      // the pivot row is in registers
      if (dpiv64) {
        Row64 r;
#pragma unroll
        for (int c = 0; c < 4; c++) r.v[c] = (64 * c + lane == pivj) ? (i64)dpiv : wneg(pr.v[c]);
        row_store64p(r, vals + (size_t)pslot * W, lane, W);
        mcw = max(mcw, lean64_publish(r, S, cst, pslot, pivj, pred, lane, nvar));
      } else {  // the denominator is no long long: that row is not one either
        i128 zw[4];
#pragma unroll
        for (int c = 0; c < 4; c++) zw[c] = (64 * c + lane == pivj) ? dpiv : -(i128)pr.v[c];
        row_store128w(zw, vals + (size_t)pslot * W, lane, W);
        mcw = max(mcw, lean64_publish_wide(zw, S, cst, pslot, pivj, pred, lane, nvar));
      }
      PROF(4);
      // two rows of the work list on their way from HBM / L2 while the one before them is updated
      Row64 rq[2];
      int sq[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; q2++) {
        sq[q2] = S.work[q2 < nwork ? q2 : 0];
        if (q2 < nwork && sq[q2] != pslot) row_load64p(rq[q2], vals + (size_t)sq[q2] * W, lane, W);
      }
      for (int w = 0; w < nwork; w++) {
        const int s = sq[0];
        Row64 r = rq[0];
        rq[0] = rq[1];
        sq[0] = sq[1];
        if (w + 2 < nwork) {
          sq[1] = S.work[w + 2];
          if (sq[1] != pslot) row_load64p(rq[1], vals + (size_t)sq[1] * W, lane, W);
        }
        if (s == pslot) continue;
        T *row = vals + (size_t)s * W;
#ifdef PIP_PROFILE
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#endif
        PROF(5);
        // multipliers from the row's own pivot-column entry (traiter.c:470-476); long longs
        i64 foo;
        {
          i64 pp = 0;
#pragma unroll
          for (int c = 0; c < 4; c++)
            if (c == pc) pp = r.v[c];
          foo = readlane64(pp, pl);
        }
        const T den_s = uni64(S.den[s]);
        i64 lp = pivot;
        T g0 = den_s;
        if (pivot != 1) {
          const u64 d = gcd_mag((u64)pivot, uabs64(foo));
          if (d != 1) {  // (d == 0 cannot be: pivot > 0)
            lp = exact_quo<i64>(pivot, (i64)d);
            foo = exact_quo<i64>(foo, (i64)d);
          }
          g0 = wmul((T)lp, den_s);
        }
        T nd;
        PROF(6);
        const T glim = (T)1 << 62;
        if (psmall && S.rcls[s] == 0 && g0 < glim && g0 > -glim) {
          // small path: every operand below 2^31, every product below 2^62; the denominator product a long long
          i64 z[4];
          u64 mx = 0;
          const i64 zf = (i64)dpiv * foo;
#pragma unroll
          for (int c = 0; c < 4; c++) {
            i64 v = r.v[c] * lp - pr.v[c] * foo;
            if (64 * c + lane == pivj) v = zf;
            z[c] = v;
            mx |= uabs64(v);
          }
          i64 nd64;
          if (!row_reduce<i64, 4>(z, mx, (i64)g0, lane, nd64, zf)) {
            if (lane == 0) sc.bad = 1;
          }
          nd = (T)nd64;
#pragma unroll
          for (int c = 0; c < 4; c++) r.v[c] = z[c];
          PROF(7);
          row_store64p(r, row, lane, W);
          mcw = max(mcw, lean64_publish(r, S, cst, s, pivj, SIG_RED, lane, nvar));
        } else {
          // mid path: long long operands, products below 2^126 -- pip_advance_kernel's update_row on the same values (its
          // wrap-around arithmetic has nothing to wrap here, except the products with denominators beyond long longs,
          // which wrap the same way)
          i128 zw[4];
          u128 mx = 0;
          const i128 zf = wmul(dpiv, (T)foo);
#pragma unroll
          for (int c = 0; c < 4; c++) {
            i128 v = (i128)r.v[c] * (i128)lp - (i128)pr.v[c] * (i128)foo;
            if (64 * c + lane == pivj) v = zf;
            zw[c] = v;
            mx |= uabs64(v);
          }
          if (!row_reduce<i128, 4>(zw, mx, g0, lane, nd, zf)) {
            if (lane == 0) sc.bad = 1;
          }
          bool fits = true;
#pragma unroll
          for (int c = 0; c < 4; c++) fits &= (uabs64(zw[c]) >> 63) == 0;
          PROF(7);
          if (ballot64(!fits) == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++) r.v[c] = (i64)zw[c];
            row_store64p(r, row, lane, W);
            mcw = max(mcw, lean64_publish(r, S, cst, s, pivj, SIG_RED, lane, nvar));
          } else {  // not a row of long longs any more: general format, the lean run ends after this pivot
            row_store128w(zw, row, lane, W);
            mcw = max(mcw, lean64_publish_wide(zw, S, cst, s, pivj, SIG_RED, lane, nvar));
          }
        }
        if (lane == 0) S.den[s] = nd;
        PROF(8);
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
        v = (T)cst[rf];  // (the constant terms are kept current in LDS by lean64_publish)
        d = S.den[rf];
      }
      sol_num[i] = v;
      sol_den[i] = d;
    }
  }
  if (status == PIPAMD_ST_RUN || status == PIPAMD_ST_CAPACITY) {
    // the job goes on elsewhere (pip_advance_kernel, pip_rehouse_kernel): its rows in the general format again
    rows_unpack64(vals, ni, lane, W, S.rcls);
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
    J->state_nch = 4;
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
      atomicAdd(q.out_maxni + 1, 1);
    }
  }
  PROF(11);
#ifdef PIP_PROFILE
  PROF_FLUSH(prof);
#endif
}

// the launch: a.Smax / a.Lmax = the row capacity of the LDS image (the caller sizes it with lean64_lds_bytes)
inline hipError_t launch_lean64(const AdvanceLaunch &a) {
  const int grid = a.grid > 0 && a.grid < a.njobs ? a.grid : a.njobs;
  const size_t shm = lean64_lds_bytes(a.Smax, a.Lmax);
  const void *fn = (const void *)pip_lean64_kernel;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (shm > 48 * 1024) {
    static std::atomic<unsigned long long> raised{0};
    if (!((raised.load(std::memory_order_acquire) >> dev) & 1)) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PIPAMD_LDS_BUDGET);
      if (e != hipSuccess) return e;
      raised.fetch_or(1ull << dev, std::memory_order_release);
    }
  }
#ifdef PIP_PROFILE
  hipLaunchKernelGGL(pip_lean64_kernel, dim3(grid), dim3(64), shm, a.stream, a.jobs, a.arena, a.njobs, a.Smax, a.Lmax, a.iter_limit,
                     a.q, (u64 *)a.prof);
#else
  hipLaunchKernelGGL(pip_lean64_kernel, dim3(grid), dim3(64), shm, a.stream, a.jobs, a.arena, a.njobs, a.Smax, a.Lmax, a.iter_limit,
                     a.q);
#endif
  return hipGetLastError();
}
#endif  // PIP_LEAN64_H
