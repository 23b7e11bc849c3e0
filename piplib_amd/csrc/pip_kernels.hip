// piplib_amd/csrc/pip_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X).
//
// One workgroup (256 threads = 4 wave64) owns one PIP problem ("job") and runs
// PipLib's dual-simplex pivot loop on it:
//
//   traiter()      reference source/traiter.c:628-791   -> pip_advance_kernel main loop
//   chercher()     traiter.c:39-44                      -> phase A (first Minus row)
//   exam_coef()    traiter.c:101-159                    -> exam_rows()   (from per-row sign summaries)
//   choisir_piv()  traiter.c:297-341                    -> choose_column() (wave 0, row-ordered tournament)
//   pivoter()      traiter.c:345-548                    -> phases C1..C5
//   integrer()     integrer.c:305-534 (constant cuts)   -> gomory_cut()
//   tab_sort_rows  traiter.c:556-623                    -> sort_rows()
//
// Data layout (all int64 "Entier" numerators, wrap-around arithmetic exactly as
// the reference's `long long` build):
//   * the tableau lives in HBM: S row slots of W int64 each (W even, 16-byte
//     aligned rows so a wave reads/writes a row with 16 B per lane, coalesced);
//   * logical row i is either a unit row (identity on column ref[i]) or a real
//     row stored in slot ref[i]; flags/denominators/ref of all logical rows are
//     staged in LDS for the whole solve, as are the pivot row, the per-row
//     multipliers derived from the pivot column, and a per-row sign summary so
//     that sign tests never touch HBM;
//   * wave-level ballots / shuffles implement the pivot-column tournament, the
//     row-gcd refinement and all sign tests.  No MFMA: exact integer work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pip_job.h"

typedef long long i64;
typedef unsigned long long u64;

#define NT 256
#define NW 4
#define BIG_I 0x7fffffff

// ---------------------------------------------------------------- integer ops
// piplib.h:128-169 + integrer.c:43-74 on wrap-around 64-bit integers.
__device__ __forceinline__ u64 uabs64(i64 x) { return x < 0 ? 0ull - (u64)x : (u64)x; }
__device__ __forceinline__ i64 wmul(i64 a, i64 b) { return (i64)((u64)a * (u64)b); }
__device__ __forceinline__ i64 wsub(i64 a, i64 b) { return (i64)((u64)a - (u64)b); }
__device__ __forceinline__ i64 wadd(i64 a, i64 b) { return (i64)((u64)a + (u64)b); }
__device__ __forceinline__ i64 wneg(i64 a) { return (i64)(0ull - (u64)a); }

// Binary gcd on magnitudes == |Euclid(a,b)| of integrer.c:43-50.
__device__ __forceinline__ u64 gcd_u64(u64 a, u64 b) {
  if (a == 0) return b;
  if (b == 0) return a;
  int sh = __builtin_ctzll(a | b);
  a >>= __builtin_ctzll(a);
  do {
    b >>= __builtin_ctzll(b);
    if (a > b) {
      u64 t = a;
      a = b;
      b = t;
    }
    b -= a;
  } while (b);
  return a << sh;
}
__device__ __forceinline__ i64 gcd_i64(i64 a, i64 b) { return (i64)gcd_u64(uabs64(a), uabs64(b)); }
// C '/' and '%' made total (the CPU traps on x / 0 and MIN / -1).
__device__ __forceinline__ i64 cquo(i64 a, i64 b) {
  if (b == 0) return 0;
  if (b == -1) return wneg(a);
  return a / b;
}
__device__ __forceinline__ i64 crem(i64 a, i64 b) {
  if (b == 0 || b == -1) return 0;
  return a % b;
}
// integrer.c:69-74 piplib_llmod
__device__ __forceinline__ i64 fmod64(i64 a, i64 b) {
  i64 m = crem(a, b);
  if (m < 0) m = wadd(m, (i64)uabs64(b));
  return m;
}
// integrer.c:51-59 piplib_lllog2
__device__ __forceinline__ int log2_64(i64 x) {
  u64 u = uabs64(x);
  int n = 64 - __builtin_clzll(u | 1ull);
  return u == 0 ? 1 : n;
}
// inverse of an odd number modulo 2^64 (Newton), for exact division
__device__ __forceinline__ u64 inv_odd64(u64 m) {
  u64 x = m;  // 3 correct bits
  x *= 2 - m * x;
  x *= 2 - m * x;
  x *= 2 - m * x;
  x *= 2 - m * x;
  x *= 2 - m * x;
  return x;
}
__device__ __forceinline__ int sign_code(i64 x) { return x > 0 ? 1 : (x < 0 ? 2 : 0); }  // 0 zero 1 plus 2 minus

__device__ __forceinline__ i64 shfl64(i64 v, int src) {
  int lo = __shfl((int)(u64)v, src), hi = __shfl((int)((u64)v >> 32), src);
  return (i64)(((u64)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  for (int o = 32; o; o >>= 1) {
    u64 t = (u64)shfl64((i64)v, (threadIdx.x & 63) ^ o);
    v = t > v ? t : v;
  }
  return v;
}

// per-row sign summary kept in LDS (exam_coef and the post-pivot flag update read only this)
//  bits 0-1 constant term, bit 2 some parameter coef > 0, bit 3 some < 0,
//  bits 4-5 big-parameter coef, bits 6-7 coef in the column just pivoted on.
#define SIG_CONST(s) ((s)&3)
#define SIG_PPOS(s) (((s) >> 2) & 1)
#define SIG_PNEG(s) (((s) >> 3) & 1)
#define SIG_BIG(s) (((s) >> 4) & 3)
#define SIG_PIV(s) (((s) >> 6) & 3)

struct Shared {
  i64 *den;    // [Lmax] denominators by logical row
  i64 *prow;   // [Wmax] pivot row (zero beyond ncol)
  i64 *lpiv;   // [Smax] per slot: pivot / gcd(pivot, foo)
  i64 *foo;    // [Smax] per slot: foo / gcd
  i64 *g0;     // [Smax] per slot: lpiv * old denominator
  int *flag;   // [Lmax]
  int *ref;    // [Lmax] slot (real row) or unit column
  int *sig;    // [Lmax]
  int *srow;   // [Smax] slot -> logical row
  int *urow;   // [Wmax] unknown column -> logical row of its unit row (or -1)
  float *size; // [Lmax] tab_sort_rows key
};

struct Scalars {
  int pivi, pivj, tmp, tmp2, status, aux;
  int flagor;
  u64 maxabs;
  i64 pivot, dpiv;
};

template <int NCH>
struct RowRegs {
  i64 v[NCH][2];
};

// ---- coalesced row access: lane l of a wave owns columns c*128 + 2l, +1 ----
template <int NCH>
__device__ __forceinline__ void row_load(RowRegs<NCH> &r, const i64 *row, int ncolp, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    int j0 = c * 128 + 2 * lane;
    if (j0 < ncolp) {
      const longlong2 t = *reinterpret_cast<const longlong2 *>(row + j0);
      r.v[c][0] = t.x;
      r.v[c][1] = t.y;
    } else {
      r.v[c][0] = 0;
      r.v[c][1] = 0;
    }
  }
}
template <int NCH>
__device__ __forceinline__ void row_store(const RowRegs<NCH> &r, i64 *row, int ncolp, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    int j0 = c * 128 + 2 * lane;
    if (j0 < ncolp) {
      longlong2 t;
      t.x = r.v[c][0];
      t.y = r.v[c][1];
      *reinterpret_cast<longlong2 *>(row + j0) = t;
    }
  }
}

// sign summary + running max|entry| of a row held in registers (wave-collective)
template <int NCH>
__device__ __forceinline__ int row_signature(const RowRegs<NCH> &r, int nvar, int ncol, int bigparm, int pivj,
                                              int lane, u64 &maxabs) {
  int cs = 0, bs = 0, ps = 0;
  bool ppos = false, pneg = false;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int j = c * 128 + 2 * lane + h;
      i64 z = r.v[c][h];
      u64 a = uabs64(z);
      maxabs = a > maxabs ? a : maxabs;
      if (j == nvar) cs = sign_code(z);
      if (j == bigparm) bs = sign_code(z);
      if (j == pivj) ps = sign_code(z);
      if (j > nvar && j < ncol) {
        ppos |= z > 0;
        pneg |= z < 0;
      }
    }
  // every field is owned by exactly one lane (or is an OR): combine with ballots
  int sig = 0;
  sig |= (__ballot(cs == 1) ? 1 : 0) | (__ballot(cs == 2) ? 2 : 0);
  sig |= (__ballot(ppos) ? 4 : 0) | (__ballot(pneg) ? 8 : 0);
  sig |= (__ballot(bs == 1) ? 16 : 0) | (__ballot(bs == 2) ? 32 : 0);
  sig |= (__ballot(ps == 1) ? 64 : 0) | (__ballot(ps == 2) ? 128 : 0);
  return sig;
}

// pivoter()'s inner loop for one row (traiter.c:470-501), one wave per row.
//   z_j = p_j*lpiv - q_j*foo  (j != pivj),  z_pivj = dpiv*foo
//   g   = gcd(lpiv*den, z_0, ..., z_{ncol-1});  row /= g; den = lpiv*den/g
// The reference folds the gcd left to right and stops calling gcd once it hits
// 1; gcd is associative, so any evaluation order gives the same g.  We refine
// g downwards: reduce every z modulo the current g, fold in one non-zero
// remainder, repeat until all remainders vanish (typically <= 2 rounds).
template <int NCH>
__device__ __forceinline__ bool update_row(RowRegs<NCH> &r, const i64 *prow, int pivj, i64 lpiv, i64 foo, i64 dpiv,
                                           i64 g0, int lane, i64 &newden) {
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int j = c * 128 + 2 * lane + h;
      i64 q = prow[j];
      i64 z = wsub(wmul(r.v[c][h], lpiv), wmul(q, foo));
      if (j == pivj) z = wmul(dpiv, foo);
      r.v[c][h] = z;
    }
  newden = g0;
  if (g0 == 1) return true;
  u64 g = uabs64(g0);
  for (;;) {
    u64 rr = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        u64 a = uabs64(r.v[c][h]);
        u64 m = g ? a % g : a;
        rr = rr ? rr : m;
      }
    u64 nz = __ballot(rr != 0);
    if (!nz) break;
    int src = __ffsll((long long)nz) - 1;
    u64 r0 = (u64)shfl64((i64)rr, src);
    g = gcd_u64(g, r0);
    if (g == 1) break;
  }
  if (g == 1) return true;
  if (g == 0) return false;  // the reference would divide by zero here
  int s = __builtin_ctzll(g);
  u64 inv = inv_odd64(g >> s);
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) r.v[c][h] = (i64)((u64)(r.v[c][h] >> s) * inv);
  newden = (i64)((u64)(g0 >> s) * inv);
  return true;
}

// ------------------------------------------------------------------ exam_coef
// traiter.c:101-159 from the LDS sign summaries.  Block-collective; returns the
// first row proven negative or BIG_I.
__device__ int exam_rows(const Shared &S, Scalars *sc, int nligne, int bigparm) {
  const int tid = threadIdx.x;
  if (bigparm >= 0) {
    if (tid == 0) sc->tmp = BIG_I;
    __syncthreads();
    for (int i = tid; i < nligne; i += NT)
      if (S.flag[i] == PIPAMD_F_UNKNOWN && SIG_BIG(S.sig[i]) == 2) atomicMin(&sc->tmp, i);
    __syncthreads();
    int i1 = sc->tmp;
    for (int i = tid; i < nligne; i += NT)
      if (S.flag[i] == PIPAMD_F_UNKNOWN) {
        if (i == i1)
          S.flag[i] = PIPAMD_F_MINUS;
        else if (i < i1 && SIG_BIG(S.sig[i]) == 1)
          S.flag[i] = PIPAMD_F_PLUS;
      }
    __syncthreads();
    if (i1 != BIG_I) return i1;
  }
  if (tid == 0) sc->tmp = BIG_I;
  __syncthreads();
  int nf[4];  // up to 1024 logical rows / 256 threads
  int cnt = 0;
  for (int i = tid; i < nligne; i += NT, cnt++) {
    int f = 0;
    if (S.flag[i] == PIPAMD_F_UNKNOWN) {
      int sg = S.sig[i];
      int fc = SIG_CONST(sg) == 1 ? PIPAMD_F_PLUS : (SIG_CONST(sg) == 2 ? PIPAMD_F_MINUS : PIPAMD_F_ZERO);
      int pp = SIG_PPOS(sg), pn = SIG_PNEG(sg);
      if (pp && pn)
        f = PIPAMD_F_UNKNOWN;
      else if (pp)
        f = (fc == PIPAMD_F_MINUS) ? PIPAMD_F_UNKNOWN : PIPAMD_F_PLUS;
      else if (pn)
        f = (fc != PIPAMD_F_MINUS) ? PIPAMD_F_UNKNOWN : PIPAMD_F_MINUS;
      else
        f = fc;
      if (f == PIPAMD_F_MINUS) atomicMin(&sc->tmp, i);
    }
    nf[cnt & 3] = f;
  }
  __syncthreads();
  int i2 = sc->tmp;
  cnt = 0;
  for (int i = tid; i < nligne; i += NT, cnt++)
    if (nf[cnt & 3] && i <= i2) S.flag[i] = nf[cnt & 3];
  __syncthreads();
  return i2;
}

// -------------------------------------------------------------- choisir_piv
// traiter.c:297-341.  The reference scans candidate columns j (positive entry
// a_j in the pivot row) and keeps the one whose column, divided by a_j, is
// lexicographically smallest over the logical rows 0..nligne-1.  We walk the
// rows once instead, keeping the set of columns still tied for the minimum:
//   * a unit row (identity on column u) is > 0 only in column u: it removes u
//     from the tied set unless u is the last one left;
//   * a real row keeps the columns with minimal v[k][j]/a_j (exact
//     cross-multiplication, ties kept);
// and stop when one column is left.  Executed by wave 0 only.
// Exact while (max a_j) * (max |entry|) < 2^62, which the caller guarantees.
template <int NCH>
__device__ int choose_column(const Shared &S, const i64 *vals, int W, int nvar, int nligne, int pivi, int ncolp,
                             Scalars *sc) {
  const int lane = threadIdx.x & 63;
  i64 a[NCH][2];
  int u[NCH][2];
  bool cand[NCH][2];
  int count = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      int j = c * 128 + 2 * lane + h;
      a[c][h] = j < nvar ? S.prow[j] : 0;
      cand[c][h] = a[c][h] > 0;
      u[c][h] = cand[c][h] ? S.urow[j] : -1;
      count += __popcll(__ballot(cand[c][h]));
    }
  if (count == 0) return -1;
  for (int k0 = 0; k0 < nligne && count > 1; k0 += 64) {
    int k = k0 + lane;
    bool real = k < nligne && !(S.flag[k] & PIPAMD_F_UNIT) && k != pivi;
    u64 realmask = __ballot(real);
    while (realmask && count > 1) {
      int kk = k0 + __ffsll((long long)realmask) - 1;
      realmask &= realmask - 1;
      // unit rows above kk knock out their own column
      int nel = 0;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int h = 0; h < 2; h++) nel += __popcll(__ballot(cand[c][h] && u[c][h] < kk));
      if (nel == count) goto last_unit_wins;
      if (nel) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++)
            if (u[c][h] < kk) cand[c][h] = false;
        count -= nel;
        if (count == 1) break;
      }
      // real row kk: keep the minimal ratios
      RowRegs<NCH> n;
      row_load<NCH>(n, vals + (size_t)S.ref[kk] * W, ncolp, lane);
      for (;;) {
        // reference column b = first remaining candidate
        i64 ab = 0, nb = 0;
        bool got = false;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            u64 m = __ballot(cand[c][h]);
            if (!got && m) {
              int src = __ffsll((long long)m) - 1;
              ab = shfl64(a[c][h], src);
              nb = shfl64(n.v[c][h], src);
              got = true;
            }
          }
        bool neg[NCH][2];
        int nneg = 0, nzero = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            i64 x = wsub(wmul(ab, n.v[c][h]), wmul(nb, a[c][h]));
            neg[c][h] = cand[c][h] && x < 0;
            bool zero = cand[c][h] && x == 0;
            nneg += __popcll(__ballot(neg[c][h]));
            nzero += __popcll(__ballot(zero));
            if (!neg[c][h] && !zero) cand[c][h] = false;  // strictly larger: out
          }
        if (nneg == 0) {
          count = nzero;
          break;
        }
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) cand[c][h] = neg[c][h];
        count = nneg;
        if (count == 1) break;
      }
    }
  }
  if (count == 1) {
    int res = -1;
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int h = 0; h < 2; h++) {
        u64 m = __ballot(cand[c][h]);
        if (m) res = c * 128 + 2 * (__ffsll((long long)m) - 1) + h;
      }
    return res;
  }
last_unit_wins:
  // only unit rows left to look at: the column whose unit row comes last survives
  if (lane == 0) sc->tmp2 = -1;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (cand[c][h]) atomicMax(&sc->tmp2, (u[c][h] << 10) | (c * 128 + 2 * lane + h));
  __builtin_amdgcn_wave_barrier();
  return sc->tmp2 & 1023;
}

// ------------------------------------------------------------ tab_sort_rows
// traiter.c:591-614: selection sort of the real rows nvar..nligne-1 by `size`
// (first minimum strictly below the running bound, swap into place).  Wave 0.
__device__ void sort_rows(const Shared &S, int nvar, int nligne, float smaxf, double smax) {
  const int lane = threadIdx.x & 63;
  for (int i = nvar; i < nligne; i++) {
    if (S.flag[i] & PIPAMD_F_UNIT) continue;
    // first argmin of size[j], j >= i, among real rows with size < smax
    float best = 0;
    int bj = BIG_I;
    for (int j = i + lane; j < nligne; j += 64) {
      if (S.flag[j] & PIPAMD_F_UNIT) continue;
      float sj = S.size[j];
      if (!((double)sj < smax)) continue;
      if (bj == BIG_I || sj < best) {
        best = sj;
        bj = j;
      }
    }
    for (int o = 32; o; o >>= 1) {
      float ob = __shfl(best, lane ^ o);
      int oj = __shfl(bj, lane ^ o);
      if (oj != BIG_I && (bj == BIG_I || ob < best || (ob == best && oj < bj))) {
        best = ob;
        bj = oj;
      }
    }
    int pv = (bj == BIG_I) ? i : bj;
    if (pv != i && lane == 0) {
      int tf = S.flag[pv], tr = S.ref[pv], tg = S.sig[pv];
      i64 td = S.den[pv];
      float ts = S.size[pv];
      S.flag[pv] = S.flag[i];
      S.ref[pv] = S.ref[i];
      S.sig[pv] = S.sig[i];
      S.den[pv] = S.den[i];
      S.size[pv] = S.size[i];
      S.flag[i] = tf;
      S.ref[i] = tr;
      S.sig[i] = tg;
      S.den[i] = td;
      S.size[i] = ts;
    }
    __builtin_amdgcn_wave_barrier();
    (void)smaxf;
  }
}

// x86 cvttsd2si semantics of the reference's (int)t, traiter.c:583
__device__ __forceinline__ int trunc_int_x86(double t) {
  if (!(t > -2147483649.0 && t < 2147483648.0)) return (int)0x80000000;
  return (int)t;
}

// ================================================================ main kernel
template <int NCH>
__global__ __launch_bounds__(NT) void pip_advance_kernel(PipJob *jobs, i64 *arena, int njobs, int Lmax, int Smax,
                                                         int Wmax, int iter_limit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ Scalars sc;
  const int jb = blockIdx.x;
  if (jb >= njobs) return;
  PipJob *J = &jobs[jb];
  if (J->status != PIPAMD_ST_RUN) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int WP = NCH * 128;  // columns a wave's registers cover; prow/urow are padded to it
  (void)Wmax;

  Shared S;
  {
    unsigned char *p = smem;
    S.den = (i64 *)p;   p += sizeof(i64) * Lmax;
    S.prow = (i64 *)p;  p += sizeof(i64) * WP;
    S.lpiv = (i64 *)p;  p += sizeof(i64) * Smax;
    S.foo = (i64 *)p;   p += sizeof(i64) * Smax;
    S.g0 = (i64 *)p;    p += sizeof(i64) * Smax;
    S.flag = (int *)p;  p += sizeof(int) * Lmax;
    S.ref = (int *)p;   p += sizeof(int) * Lmax;
    S.sig = (int *)p;   p += sizeof(int) * Lmax;
    S.srow = (int *)p;  p += sizeof(int) * Smax;
    S.urow = (int *)p;  p += sizeof(int) * WP;
    S.size = (float *)p;
  }

  const int nvar = J->nvar, nparm = J->nparm, bigparm = J->bigparm;
  int tflags = J->tflags;
  int ni = J->ni;
  const int L = J->L, Sl = J->S, W = J->W;
  const int ncol = nvar + nparm + 1;
  const int ncolp = (ncol + 1) & ~1;
  i64 *vals = arena + J->vals_off;
  i64 *g_den = arena + J->rows_off;
  int *g_flag = (int *)(g_den + L);
  int *g_ref = g_flag + L;
  int nligne = nvar + ni;
  int npiv = J->npiv, ncut = J->ncut;
  int ldet = J->ldet;
  i64 det[PIPAMD_MAXDET];
  for (int i = 0; i < PIPAMD_MAXDET; i++) det[i] = J->det[i];

  // ---- stage the row tables in LDS -------------------------------------
  for (int i = tid; i < nligne; i += NT) {
    S.den[i] = g_den[i];
    S.flag[i] = g_flag[i];
    S.ref[i] = g_ref[i];
    S.sig[i] = 0;
    S.size[i] = 0.f;
  }
  for (int j = tid; j < WP; j += NT) {
    S.urow[j] = -1;
    S.prow[j] = 0;
  }
  if (tid == 0) {
    sc.maxabs = 0;
    sc.status = PIPAMD_ST_RUN;
    sc.aux = 0;
  }
  __syncthreads();
  for (int i = tid; i < nligne; i += NT) {
    if (S.flag[i] & PIPAMD_F_UNIT)
      S.urow[S.ref[i]] = i;
    else
      S.srow[S.ref[i]] = i;
  }
  __syncthreads();
  // ---- one pass over the tableau: sign summaries, max |entry|, sort keys --
  {
    u64 mx = 0;
    for (int s = wave; s < ni; s += NW) {
      RowRegs<NCH> r;
      int k = S.srow[s];
      row_load<NCH>(r, vals + (size_t)s * W, ncolp, lane);
      int sg = row_signature<NCH>(r, nvar, ncol, bigparm, -1, lane, mx);
      if (tflags & PIPAMD_T_SORT) {
        // traiter.c:576-589: size = max_j |(int)(v_j / den)| over the unknowns
        double d = (double)S.den[k], sz = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            int j = c * 128 + 2 * lane + h;
            if (j < nvar) {
              int q = trunc_int_x86((double)r.v[c][h] / d);
              double aq = (double)(q < 0 ? (int)(0u - (unsigned)q) : q);
              sz = sz > aq ? sz : aq;
            }
          }
        for (int o = 32; o; o >>= 1) {
          double t = __shfl(sz, lane ^ o);
          sz = sz > t ? sz : t;
        }
        if (lane == 0) S.size[k] = (float)sz;
        if (lane == 0) S.lpiv[s] = (i64)__double_as_longlong(sz);  // exact double key for smax
      }
      if (lane == 0) S.sig[k] = sg;
    }
    mx = wave_max_u64(mx);
    if (lane == 0) atomicMax(&sc.maxabs, mx);
  }
  __syncthreads();
  if (tflags & PIPAMD_T_SORT) {
    if (wave == 0) {
      // smax over rows nvar..nligne-1 only (traiter.c:576-586)
      double smax = 0;
      for (int i = nvar + lane; i < nligne; i += 64)
        if (!(S.flag[i] & PIPAMD_F_UNIT)) {
          double t = __longlong_as_double(S.lpiv[S.ref[i]]);
          smax = smax > t ? smax : t;
        }
      for (int o = 32; o; o >>= 1) {
        double t = __shfl(smax, lane ^ o);
        smax = smax > t ? smax : t;
      }
      sort_rows(S, nvar, nligne, 0.f, smax);
    }
    __syncthreads();
    for (int i = tid; i < nligne; i += NT)
      if (!(S.flag[i] & PIPAMD_F_UNIT)) S.srow[S.ref[i]] = i;
    tflags &= ~PIPAMD_T_SORT;
    __syncthreads();
  }

  int status = PIPAMD_ST_RUN;
  for (int iter = 0;; iter++) {
    if (iter >= iter_limit) break;  // status stays RUN: the host relaunches
    // ---------------- A: chercher(Minus), then exam_coef ------------------
    if (tid == 0) {
      sc.pivi = BIG_I;
      sc.flagor = 0;
    }
    __syncthreads();
    for (int i = tid; i < nligne; i += NT)
      if (S.flag[i] & PIPAMD_F_MINUS) atomicMin(&sc.pivi, i);
    __syncthreads();
    int pivi = sc.pivi;
    if (pivi == BIG_I) {
      pivi = exam_rows(S, &sc, nligne, bigparm);
      if (pivi == BIG_I) {
        if (nparm > 0) {
          for (int i = tid; i < nligne; i += NT)
            if (S.flag[i] & (PIPAMD_F_CRITIC | PIPAMD_F_UNKNOWN)) atomicOr(&sc.flagor, 1);
          __syncthreads();
          if (sc.flagor) {
            status = PIPAMD_ST_NEED_COMPA;
            break;
          }
        }
        if (!(tflags & PIPAMD_T_INT)) {
          status = PIPAMD_ST_SOLUTION;
          break;
        }
        // ------------- integrer(): first non-integral row among the unknowns
        if (ncol >= PIPAMD_MAXCOL) {
          status = PIPAMD_ST_MAXCOL;
          break;
        }
        if (tid == 0) sc.tmp = BIG_I;
        __syncthreads();
        for (int i = tid; i < nvar; i += NT) {
          i64 D = S.den[i];
          if (D == 1 || (S.flag[i] & PIPAMD_F_UNIT)) continue;
          const i64 *row = vals + (size_t)S.ref[i] * W;
          bool ok = wneg(fmod64(wneg(row[nvar]), D)) != 0;
          for (int j = nvar + 1; j < ncol && !ok; j++)
            if (j != bigparm && fmod64(wneg(row[j]), D) != 0) ok = true;
          if (ok) atomicMin(&sc.tmp, i);
        }
        __syncthreads();
        int ci = sc.tmp;
        if (ci == BIG_I) {
          status = PIPAMD_ST_SOLUTION;
          break;
        }
        // build the cut in the pivot-row buffer (integrer.c:357-386)
        {
          const i64 D = S.den[ci];
          const i64 *row = vals + (size_t)S.ref[ci] * W;
          int okv = 0, okp = 0;
          for (int j = tid; j < WP; j += NT) {
            i64 x = 0;
            if (j < ncol) {
              i64 v = row[j];
              if (j < nvar) {
                x = fmod64(v, D);
                okv |= x > 0;
              } else if (j == nvar) {
                x = wneg(fmod64(wneg(v), D));
              } else if (j != bigparm) {
                x = wneg(fmod64(wneg(v), D));
                okp |= x != 0;
              }
            }
            S.prow[j] = x;
          }
          if (okv) atomicOr(&sc.flagor, 2);
          if (okp) atomicOr(&sc.flagor, 4);
          __syncthreads();
          int fo = sc.flagor;
          if (fo & 4) {  // parametric cut: the host owns the context (find_parm/add_parm)
            status = PIPAMD_ST_NEED_PARMCUT;
            if (tid == 0) sc.aux = ci;
            break;
          }
          if (!(fo & 2)) {  // integrer.c:482-485 case (b)
            status = PIPAMD_ST_NIL;
            break;
          }
          if (tflags & PIPAMD_T_DEEPEST) {
            status = PIPAMD_ST_INTERNAL;  // deepest cut is applied by the host path only
            break;
          }
          if (ni >= Sl || nligne >= L) {
            status = PIPAMD_ST_CAPACITY;
            break;
          }
          // append the cut as logical row nligne in slot ni (integrer.c:440-446)
          i64 *nrow = vals + (size_t)ni * W;
          u64 mx = 0;
          for (int j = tid; j < ncolp; j += NT) {
            nrow[j] = S.prow[j];
            u64 a = uabs64(S.prow[j]);
            mx = a > mx ? a : mx;
          }
          mx = wave_max_u64(mx);
          if (lane == 0) atomicMax(&sc.maxabs, mx);
          if (tid == 0) {
            S.flag[nligne] = PIPAMD_F_MINUS;
            S.den[nligne] = D;
            S.ref[nligne] = ni;
            S.srow[ni] = nligne;
            S.sig[nligne] = sign_code(S.prow[nvar]);  // parameters are all zero here
            S.size[nligne] = 0.f;
          }
          pivi = nligne;
          ni++;
          nligne++;
          ncut++;
          __syncthreads();
        }
      }
    }
    // ---------------- C1: stage the pivot row ------------------------------
    npiv++;
    const int pslot = S.ref[pivi];
    {
      const i64 *row = vals + (size_t)pslot * W;
      for (int j = tid; j < WP; j += NT) S.prow[j] = j < ncol ? row[j] : 0;
    }
    __syncthreads();
    // ---------------- C2: choisir_piv (wave 0) ------------------------------
    if (wave == 0) {
      // exactness guard of the tournament: (max candidate a_j) * (max |entry|) < 2^62
      u64 amax = 0;
      for (int j = lane; j < nvar; j += 64) {
        i64 a = S.prow[j];
        if (a > 0 && (u64)a > amax) amax = (u64)a;
      }
      amax = wave_max_u64(amax);
      u64 mx = sc.maxabs;
      bool safe = amax == 0 || mx == 0 || (__umul64hi(amax, mx) == 0 && amax * mx < (1ull << 62));
      int pj = safe ? choose_column<NCH>(S, vals, W, nvar, nligne, pivi, ncolp, &sc) : -2;
      if (lane == 0) sc.pivj = pj;
    }
    __syncthreads();
    const int pivj = sc.pivj;
    if (pivj == -1) {  // traiter.c:782-785
      status = PIPAMD_ST_NIL;
      break;
    }
    if (pivj == -2) {
      status = PIPAMD_ST_RANGE;
      break;
    }
    // ---------------- C3: pivot scalars + per-row multipliers ---------------
    const i64 pivot = S.prow[pivj];
    const i64 dpiv = S.den[pivi];
    if (tid == 0) {
      // determinant bookkeeping, traiter.c:394-446 (every thread keeps det[] in
      // registers identically; thread 0 publishes the verdict)
      sc.tmp = 0;
    }
    {
      i64 d = gcd_i64(pivot, dpiv);
      i64 ppivot = cquo(pivot, d), dppiv = cquo(dpiv, d);
      for (int i = 0; i < ldet; i++) {
        d = gcd_i64(det[i], dppiv);
        det[i] = cquo(det[i], d);
        dppiv = cquo(dppiv, d);
      }
      bool ovf = dppiv != 1;
      if (!ovf) {
        int i = 0;
        for (; i < ldet; i++)
          if (log2_64(det[i]) + log2_64(ppivot) < 64) {
            det[i] = wmul(det[i], ppivot);
            break;
          }
        if (i >= ldet) {
          ldet++;
          if (ldet >= PIPAMD_MAXDET)
            ovf = true;
          else
            det[i] = ppivot;
        }
      }
      if (ovf) {
        status = PIPAMD_ST_OVERFLOW;
        break;
      }
    }
    for (int s = tid; s < ni; s += NT) {
      if (s == pslot) continue;
      int k = S.srow[s];
      i64 foo = vals[(size_t)s * W + pivj];
      i64 d = gcd_i64(pivot, foo);
      i64 lp = cquo(pivot, d);
      S.lpiv[s] = lp;
      S.foo[s] = cquo(foo, d);
      S.g0[s] = wmul(lp, S.den[k]);
    }
    const int ku = S.urow[pivj];  // unit row of the entering column
    if (tid == 0) sc.maxabs = 0;
    __syncthreads();
    // ---------------- C4: eliminate the pivot column from every real row ----
    {
      u64 mx = 0;
      bool bad = false;
      for (int s = wave; s < ni; s += NW) {
        RowRegs<NCH> r;
        i64 *row = vals + (size_t)s * W;
        if (s == pslot) {
          // the slot is recycled for the row replacing ku's unit row (traiter.c:461-465,503-513)
#pragma unroll
          for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              int j = c * 128 + 2 * lane + h;
              r.v[c][h] = (j == pivj) ? dpiv : wneg(S.prow[j]);
            }
          row_store<NCH>(r, row, ncolp, lane);
          int sg = row_signature<NCH>(r, nvar, ncol, bigparm, pivj, lane, mx);
          if (lane == 0) S.sig[ku] = sg;
        } else {
          int k = S.srow[s];
          i64 nd;
          row_load<NCH>(r, row, ncolp, lane);
          if (!update_row<NCH>(r, S.prow, pivj, S.lpiv[s], S.foo[s], dpiv, S.g0[s], lane, nd)) bad = true;
          row_store<NCH>(r, row, ncolp, lane);
          int sg = row_signature<NCH>(r, nvar, ncol, bigparm, pivj, lane, mx);
          if (lane == 0) {
            S.sig[k] = sg;
            S.den[k] = nd;
          }
        }
      }
      mx = wave_max_u64(mx);
      if (lane == 0) atomicMax(&sc.maxabs, mx);
      if (bad && lane == 0) atomicOr(&sc.tmp, 1);
    }
    __syncthreads();
    if (sc.tmp) {
      status = PIPAMD_ST_OVERFLOW;
      break;
    }
    // ---------------- C5: swap roles, refresh the sign hints -----------------
    if (tid == 0) {
      S.flag[ku] = PIPAMD_F_PLUS;
      S.den[ku] = pivot;
      S.ref[ku] = pslot;
      S.srow[pslot] = ku;
      S.flag[pivi] = PIPAMD_F_UNIT | PIPAMD_F_ZERO;
      S.den[pivi] = 1;
      S.ref[pivi] = pivj;
      S.urow[pivj] = pivi;
    }
    __syncthreads();
    for (int i = tid; i < nligne; i += NT) {  // traiter.c:518-529
      int ff = S.flag[i];
      if (ff & PIPAMD_F_UNIT) continue;
      int ps = SIG_PIV(S.sig[i]);
      int fff = ps == 1 ? PIPAMD_F_PLUS : (ps == 2 ? PIPAMD_F_MINUS : PIPAMD_F_ZERO);
      if (fff != PIPAMD_F_ZERO && fff != ff) {
        if (ff == PIPAMD_F_ZERO)
          ff = (fff == PIPAMD_F_MINUS) ? PIPAMD_F_UNKNOWN : fff;
        else
          ff = PIPAMD_F_UNKNOWN;
      }
      S.flag[i] = ff;
    }
    __syncthreads();
  }

  // ---- epilogue: publish the row tables, the header and (if any) the solution
  __syncthreads();
  for (int i = tid; i < nligne; i += NT) {
    g_den[i] = S.den[i];
    g_flag[i] = S.flag[i];
    g_ref[i] = S.ref[i];
  }
  if (status == PIPAMD_ST_SOLUTION) {
    // solution(), traiter.c:255-271: rows 0..nvar-1, parameters then constant
    i64 *sol_num = arena + J->sol_off;
    i64 *sol_den = sol_num + (size_t)nvar * (nparm + 1);
    for (int e = tid; e < nvar * (nparm + 1); e += NT) {
      int i = e / (nparm + 1), jj = e % (nparm + 1);
      int col = jj < nparm ? nvar + 1 + jj : nvar;
      i64 v = 0;
      if (!(S.flag[i] & PIPAMD_F_UNIT)) v = vals[(size_t)S.ref[i] * W + col];
      sol_num[e] = v;
    }
    for (int i = tid; i < nvar; i += NT) sol_den[i] = S.den[i];
  }
  if (tid == 0) {
    J->ni = ni;
    J->npiv = npiv;
    J->ncut = ncut;
    J->ldet = ldet;
    for (int i = 0; i < PIPAMD_MAXDET; i++) J->det[i] = det[i];
    J->tflags = tflags;
    J->maxabs = sc.maxabs;
    J->aux = sc.aux;
    J->status = status;
  }
}

// ---------------------------------------------------------------- batch load
// tab_alloc + tab_get (tab.c:158-248) for a uniform batch: nvar unit rows, then
// ni Unknown rows with denominator 1; spare slots and columns zeroed.
__global__ void pip_batch_load_kernel(PipJob *jobs, i64 *arena, const i64 *rows, PipBatchLayout lay) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int ncol = lay.nvar + lay.nparm + 1;
  PipJob *J = &jobs[b];
  const int64_t base = lay.arena_off + (int64_t)b * lay.per_job;
  i64 *g_den = arena + base;
  int *g_flag = (int *)(g_den + lay.L);
  int *g_ref = g_flag + lay.L;
  i64 *vals = arena + base + 2 * (int64_t)lay.L;
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
  const i64 *src = rows + (size_t)b * lay.ni * ncol;
  for (int e = tid; e < lay.S * lay.W; e += blockDim.x) {
    int s = e / lay.W, j = e % lay.W;
    vals[e] = (s < lay.ni && j < ncol) ? src[(size_t)s * ncol + j] : 0;
  }
  if (tid == 0) {
    J->rows_off = base;
    J->vals_off = base + 2 * (int64_t)lay.L;
    J->sol_off = base + 2 * (int64_t)lay.L + (int64_t)lay.S * lay.W;
    J->nvar = lay.nvar;
    J->nparm = lay.nparm;
    J->ni = lay.ni;
    J->bigparm = lay.bigparm;
    J->tflags = lay.tflags | PIPAMD_T_SORT;
    J->L = lay.L;
    J->S = lay.S;
    J->W = lay.W;
    J->status = PIPAMD_ST_RUN;
    J->aux = 0;
    J->npiv = 0;
    J->ncut = 0;
    J->ldet = 1;
    J->det[0] = 1;
    J->det[1] = J->det[2] = J->det[3] = 0;
    J->maxabs = 0;
  }
}

__global__ void pip_batch_results_kernel(const PipJob *jobs, const i64 *arena, int njobs, int nvar, int nparm,
                                         int *status, int *pivots, int *cuts, i64 *sol_num, i64 *sol_den) {
  const int b = blockIdx.x;
  const PipJob *J = &jobs[b];
  if (threadIdx.x == 0) {
    if (status) status[b] = J->status;
    if (pivots) pivots[b] = J->npiv;
    if (cuts) cuts[b] = J->ncut;
  }
  const int nn = nvar * (nparm + 1);
  const i64 *sn = arena + J->sol_off;
  const bool ok = J->status == PIPAMD_ST_SOLUTION;
  if (sol_num)
    for (int e = threadIdx.x; e < nn; e += blockDim.x) sol_num[(size_t)b * nn + e] = ok ? sn[e] : 0;
  if (sol_den)
    for (int i = threadIdx.x; i < nvar; i += blockDim.x) sol_den[(size_t)b * nvar + i] = ok ? sn[nn + i] : 0;
}

// ------------------------------------------------------------------ launchers
extern "C" hipError_t pipk_launch_advance(PipJob *jobs, i64 *arena, int njobs, int Lmax, int Smax, int Wmax,
                                          int iter_limit, hipStream_t stream) {
  if (njobs <= 0) return hipSuccess;
  const size_t WP = Wmax <= 128 ? 128 : (Wmax <= 256 ? 256 : 512);
  size_t shm = sizeof(i64) * ((size_t)Lmax + WP + 3 * (size_t)Smax) +
               sizeof(int) * (3 * (size_t)Lmax + Smax + WP) + sizeof(float) * (size_t)Lmax;
  shm = (shm + 15) & ~(size_t)15;
  dim3 grid(njobs), block(NT);
  if (Wmax <= 128) {
    hipLaunchKernelGGL(pip_advance_kernel<1>, grid, block, shm, stream, jobs, arena, njobs, Lmax, Smax, Wmax,
                       iter_limit);
  } else if (Wmax <= 256) {
    hipLaunchKernelGGL(pip_advance_kernel<2>, grid, block, shm, stream, jobs, arena, njobs, Lmax, Smax, Wmax,
                       iter_limit);
  } else if (Wmax <= 512) {
    hipLaunchKernelGGL(pip_advance_kernel<4>, grid, block, shm, stream, jobs, arena, njobs, Lmax, Smax, Wmax,
                       iter_limit);
  } else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

extern "C" hipError_t pipk_launch_batch_load(PipJob *jobs, i64 *arena, const i64 *rows, PipBatchLayout lay,
                                             hipStream_t stream) {
  hipLaunchKernelGGL(pip_batch_load_kernel, dim3(lay.batch), dim3(256), 0, stream, jobs, arena, rows, lay);
  return hipGetLastError();
}

extern "C" hipError_t pipk_launch_batch_results(const PipJob *jobs, const i64 *arena, int njobs, int nvar, int nparm,
                                                int *status, int *pivots, int *cuts, i64 *sol_num, i64 *sol_den,
                                                hipStream_t stream) {
  hipLaunchKernelGGL(pip_batch_results_kernel, dim3(njobs), dim3(256), 0, stream, jobs, arena, njobs, nvar, nparm,
                     status, pivots, cuts, sol_num, sol_den);
  return hipGetLastError();
}
