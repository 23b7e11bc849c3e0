// piplib_amd/csrc/pip_quast.hip -- traiter() with its quast decision tree on the device, for SMALL
// parametric problems (at most 64 columns and 128 real rows), in both entry widths.
//
// One wave64 per problem runs the whole call tree of traiter() (traiter.c:628-791) without a host
// round trip: the dual simplex with lexicographic pivoting (pivoter, traiter.c:345-548), exam_coef
// (traiter.c:101-159), compa_test with its two integer feasibility sub-problems per undecided row
// (traiter.c:162-243), the forks of the quast (traiter.c:695-759; the "else" state waits on a stack in
// HBM while the "then" branch runs), Gomory cuts with new parameters (integrer.c:156-291,305-534)
// and the solution tape (sol.c:104-209).  The tableaux live in LDS.  Column work (pivot-column tournament, cut
// vectors, tape cells) has a lane per column; row work (multipliers, elimination, row gcd, exact division,
// sign tests, the selection sort of tab_sort_rows) has a lane per row.
//
// The kernel computes on true integers: every product and sum is checked, and a problem in which a
// 64-bit operation overflows (where the reference's `long long` build wraps or exits with "Integer
// overflow"), or that outgrows the reserved rows / columns / stack / tape, ends with Q_FALLBACK and
// is solved again by the host tree over pip_advance_kernel, which reproduces those cases bit for
// bit.  Whatever this kernel does finish is, cell for cell, the reference's tape.  Every loop is bounded (the
// pivot-column tournament too: overflowed products compare as garbage) and a problem's wave gives up after two
// seconds at the latest.
#include <hip/hip_runtime.h>

#include "pip_job.h"
#include "pip_quast.h"

namespace {
typedef long long w64;            // a raw 64-bit word / an input coefficient (inputs are long longs in either flavour)
typedef unsigned long long u64;   // ballot masks, bit sets
typedef __int128 w128;
typedef unsigned __int128 u128;

// ---- what the kernel needs from its entry type QI (long long: the reference's int64 build; __int128: the overflow-safe
// flavour, piplib.h:42-88), by overloading.  Everything the compiler would turn into a library call for 128 bits
// (multiplication overflow, division, conversion to double) is spelled out.
template <class QI> struct QT;
template <> struct QT<w64> { typedef u64 U; };
template <> struct QT<w128> { typedef u128 U; };
__device__ __forceinline__ w64 qshfl(w64 x, int src) { return __shfl(x, src); }
__device__ __forceinline__ w128 qshfl(w128 x, int src) {
  const u64 lo = (u64)__shfl((w64)(u64)(u128)x, src), hi = (u64)__shfl((w64)(u64)((u128)x >> 64), src);
  return (w128)(((u128)hi << 64) | lo);
}
__device__ __forceinline__ w64 qrdlane(w64 x, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)(u64)x, l), hi = __builtin_amdgcn_readlane((unsigned)((u64)x >> 32), l);
  return (w64)(((u64)hi << 32) | lo);
}
__device__ __forceinline__ w128 qrdlane(w128 x, int l) {
  const u64 lo = (u64)qrdlane((w64)(u64)(u128)x, l), hi = (u64)qrdlane((w64)(u64)((u128)x >> 64), l);
  return (w128)(((u128)hi << 64) | lo);
}
__device__ __forceinline__ int qbits(u64 u) { return u ? 64 - __clzll((long long)u) : 0; }
__device__ __forceinline__ int qbits(u128 u) { return (u64)(u >> 64) ? 128 - __clzll((long long)(u64)(u >> 64)) : qbits((u64)u); }
__device__ __forceinline__ int qctz(u64 u) { return __builtin_ctzll(u | (1ull << 63)); }
__device__ __forceinline__ int qctz(u128 u) { return (u64)u ? __builtin_ctzll((u64)u) : 64 + __builtin_ctzll((u64)(u >> 64) | (1ull << 63)); }
__device__ __forceinline__ bool qaddo(w64 a, w64 b, w64 *r) { return __builtin_add_overflow(a, b, r); }
__device__ __forceinline__ bool qsubo(w64 a, w64 b, w64 *r) { return __builtin_sub_overflow(a, b, r); }
__device__ __forceinline__ bool qmulo(w64 a, w64 b, w64 *r) { return __builtin_mul_overflow(a, b, r); }
__device__ __forceinline__ bool qaddo(w128 a, w128 b, w128 *r) { return __builtin_add_overflow(a, b, r); }
__device__ __forceinline__ bool qsubo(w128 a, w128 b, w128 *r) { return __builtin_sub_overflow(a, b, r); }
// 128 x 128: exact when both fit long longs; else by the bit lengths -- a sum of at most 126 cannot overflow, 128 and more
// must, and 127 (either) is reported as overflow: the problem is handed back, never answered wrongly
__device__ __forceinline__ bool qmulo(w128 a, w128 b, w128 *r) {
  *r = (w128)((u128)a * (u128)b);
  if ((w128)(w64)a == a && (w128)(w64)b == b) return false;
  if (a == 0 || b == 0) return false;
  const u128 ua = a < 0 ? (u128)0 - (u128)a : (u128)a, ub = b < 0 ? (u128)0 - (u128)b : (u128)b;
  return qbits(ua) + qbits(ub) > 126;
}
// 64-bit division is a ~150-instruction sequence: one copy each, out of line (operands rarely need it)
__device__ __noinline__ u64 umod_wide(u64 a, u64 b) { return a % b; }
__device__ __noinline__ u64 udiv_wide(u64 a, u64 b) { return a / b; }
__device__ __forceinline__ u64 qumod(u64 a, u64 b) { return ((a | b) >> 32) ? umod_wide(a, b) : (u64)((unsigned)a % (unsigned)b); }
__device__ __forceinline__ u64 qudiv(u64 a, u64 b) { return ((a | b) >> 32) ? udiv_wide(a, b) : (u64)((unsigned)a / (unsigned)b); }
// 128 / 128 -> quotient and remainder, shift and subtract over the difference of the bit lengths (operands that fit 64
// bits take the 64-bit routines)
__device__ __noinline__ u128 udivmod_wide(u128 a, u128 b, u128 *rem) {
  u128 q = 0;
  if (b != 0 && a >= b) {
    int sh = qbits(a) - qbits(b);
    u128 d = b << sh;
    for (; sh >= 0; sh--) {
      q <<= 1;
      if (a >= d) {
        a -= d;
        q |= 1;
      }
      d >>= 1;
    }
  }
  *rem = a;
  return q;
}
__device__ __forceinline__ u128 qumod(u128 a, u128 b) {
  if (((a | b) >> 64) == 0) return (u128)qumod((u64)a, (u64)b);
  u128 r;
  udivmod_wide(a, b, &r);
  return r;
}
__device__ __forceinline__ u128 qudiv(u128 a, u128 b) {
  if (((a | b) >> 64) == 0) return (u128)qudiv((u64)a, (u64)b);
  u128 r;
  return udivmod_wide(a, b, &r);
}
__device__ __forceinline__ double qdouble(w64 x) { return (double)x; }
__device__ __forceinline__ double qdouble(w128 x) {
  const u128 m = x < 0 ? (u128)0 - (u128)x : (u128)x;  // via the magnitude: hi*2^64 + lo on a negative value would cancel
  const double d = (double)(u64)(m >> 64) * 18446744073709551616.0 + (double)(u64)m;
  return x < 0 ? -d : d;
}
// inverse of an odd number modulo 2^64 / 2^128 (Newton: 3 correct bits, doubled by every step)
__device__ __forceinline__ u64 qinv(u64 od) {
  u64 inv = od;
#pragma unroll
  for (int q = 0; q < 5; q++) inv *= 2 - od * inv;
  return inv;
}
__device__ __forceinline__ u128 qinv(u128 od) {
  u128 inv = (u128)qinv((u64)od);
  inv *= 2 - od * inv;
  return inv;
}

enum { F_UNIT = 1, F_PLUS = 2, F_MINUS = 4, F_ZERO = 8, F_CRITIC = 16, F_UNKNOWN = 32 };
enum { C_NIL = 1, C_IF = 2, C_LIST = 3, C_FORM = 4, C_NEW = 5, C_DIV = 6, C_VAL = 7 };
enum { MAXDET = 4 };  // tab.h:67

// Every tableau lives in LDS: pointers carry the address space, so that the out-of-line functions
// below get ds_read / ds_write instead of flat accesses (a generic pointer argument costs several times
// the latency).
#define LDS __attribute__((address_space(3)))
typedef LDS int lint;

// The kernel's functions as static members of a class template over the entry type: inside it `i64` IS the entry type
// (and `uE` its unsigned twin), so the code below reads as it did when there was one flavour.
template <class QI>
struct QK {
typedef QI i64;
typedef typename QT<QI>::U uE;
typedef LDS i64 li64;

// one tableau in LDS: logical rows (unit row on column `ref`, or real row in slot `ref`)
struct Tab {
  li64 *den;   // [rows]
  li64 *val;   // [slots][W]
  lint *flag;  // [rows]
  lint *ref;   // [rows]
  lint *ldet;  // -> number of determinant limbs in use
  li64 *det;   // -> MAXDET limbs
  int W, rows_cap, slots_cap;
};

struct Wv {
  int lane;
  int bad;  // sticky: an overflow or a capacity limit was hit (uniform when tested)
  int pivots;
  int deepest;
};

// scalars of the main tableau that travel with a stack frame
struct QState {
  int nvar, nparm, ni, nc, pivi, ldet, ni0, pad1;
  i64 det[MAXDET];
  // Compute_dual: tab_sort_rows' `pos` of the traiter() call in progress (traiter.c:567-620): logical row of each of
  // the call's ni0 inequalities after its sort.  Part of the image, so a fork's frame keeps the caller's.
  unsigned short pos[64];
};

#define BAD(w) (__any((w).bad) != 0)
static __device__ __forceinline__ void wsync() { __syncthreads(); }  // one wave per workgroup: orders its LDS traffic
static __device__ __forceinline__ i64 bcast(i64 x, int src) { return qshfl(x, src); }
static __device__ __forceinline__ int popc64(u64 m) { return __popcll(m); }
static __device__ __forceinline__ int first64(u64 m) { return __ffsll((long long)m) - 1; }

static __device__ __forceinline__ i64 cmul(i64 a, i64 b, int &bad) {
  i64 r;
  bad |= qmulo(a, b, &r);
  return r;
}
static __device__ __forceinline__ i64 cadd(i64 a, i64 b, int &bad) {
  i64 r;
  bad |= qaddo(a, b, &r);
  return r;
}
static __device__ __forceinline__ i64 csub(i64 a, i64 b, int &bad) {
  i64 r;
  bad |= qsubo(a, b, &r);
  return r;
}
static __device__ __forceinline__ i64 cneg(i64 a, int &bad) { return csub(0, a, bad); }
static __device__ __forceinline__ bool fits32(i64 a) { return a == (i64)(int)a; }
static __device__ __forceinline__ i64 mul32(i64 a, i64 b) { return (i64)((w64)(int)a * (w64)(int)b); }
static __device__ __forceinline__ uE uabs(i64 a) { return a < 0 ? (uE)0 - (uE)a : (uE)a; }
static __device__ __forceinline__ uE umod(uE a, uE b) { return qumod(a, b); }
static __device__ __forceinline__ uE udiv(uE a, uE b) { return qudiv(a, b); }
// integrer.c:43-50 on true integers: gcd(|a|, |b|)
static __device__ __noinline__ uE gcd_loop(uE x, uE y) {
  while (y) {
    const uE t = umod(x, y);
    x = y;
    y = t;
  }
  return x;
}
static __device__ __forceinline__ i64 gcd64(i64 a, i64 b) {
  const uE x = uabs(a), y = uabs(b);
  if (x == 1 || y == 1) return 1;
  if (y == 0) return (i64)x;
  return (i64)gcd_loop(x, y);
}
// C '/' and '%' (truncating) for a non-zero divisor
static __device__ __forceinline__ i64 quo(i64 a, i64 b) {
  const uE q = udiv(uabs(a), uabs(b));
  return ((a < 0) != (b < 0)) ? -(i64)q : (i64)q;
}
static __device__ __forceinline__ i64 rem(i64 a, i64 b) {
  const uE r = umod(uabs(a), uabs(b));
  return a < 0 ? -(i64)r : (i64)r;
}
// integrer.c:69-74: remainder in [0, |b|)
static __device__ __forceinline__ i64 pmod(i64 a, i64 b) {
  i64 m = rem(a, b);
  if (m < 0) m += (i64)uabs(b);
  return m;
}
// piplib.h:147-149
static __device__ __forceinline__ i64 floordiv(i64 a, i64 b, int &bad) { return quo(csub(a, pmod(a, b), bad), b); }
// integrer.c:51-59: bit length of |x|, 1 for 0
static __device__ __forceinline__ int blen(i64 x) {
  const uE u = uabs(x);
  return u ? qbits(u) : 1;
}
static __device__ __forceinline__ int sgn_flag(i64 x) { return x < 0 ? F_MINUS : (x > 0 ? F_PLUS : F_ZERO); }

static __device__ __forceinline__ int wave_max_i(int x) {
  for (int o = 32; o; o >>= 1) {
    const int y = __shfl_xor(x, o);
    x = x > y ? x : y;
  }
  return x;
}
static __device__ __forceinline__ int wave_min_i(int x) {
  for (int o = 32; o; o >>= 1) {
    const int y = __shfl_xor(x, o);
    x = x < y ? x : y;
  }
  return x;
}
static __device__ __forceinline__ float wave_min_f(float x) {
  for (int o = 32; o; o >>= 1) {
    const float y = __shfl_xor(x, o);
    x = x < y ? x : y;
  }
  return x;
}

// value of logical row k in column `lane` (traiter.c:246-252 valeur); 0 beyond ncol
static __device__ __forceinline__ i64 row_at(const Tab &t, int k, int lane, int ncol) {
  const int fl = t.flag[k], rf = t.ref[k];
  if (fl & F_UNIT) return rf == lane ? t.den[k] : 0;
  return lane < ncol ? t.val[rf * t.W + lane] : 0;
}

// traiter.c:39-44 chercher: first row among 0..n-1 whose flag meets `mask`, n if none
static __device__ __forceinline__ int first_flagged(const Tab &t, int mask, int n, int lane) {
  for (int base = 0; base < n; base += 64) {
    const int k = base + lane;
    const u64 m = __ballot(k < n && (t.flag[k] & mask));
    if (m) return base + first64(m);
  }
  return n;
}

// traiter.c:101-159 exam_coef: obvious signs of Unknown rows; stops at the first row proven negative
static __device__ __noinline__ int classify_rows(Tab t, int nvar, int ncol, int bigparm, int nligne, int lane) {
  if (bigparm >= 0) {
    for (int base = 0; base < nligne; base += 64) {
      const int k = base + lane;
      int s = 0;
      if (k < nligne && t.flag[k] == F_UNKNOWN) {
        const i64 v = t.val[t.ref[k] * t.W + bigparm];
        s = v < 0 ? -1 : (v > 0 ? 1 : 0);
      }
      const u64 neg = __ballot(s < 0);
      const int stop = neg ? first64(neg) : 64;
      if (s > 0 && lane < stop) t.flag[k] = F_PLUS;
      if (neg) {
        if (lane == stop) t.flag[k] = F_MINUS;
        wsync();
        return base + stop;
      }
    }
    wsync();
  }
  for (int base = 0; base < nligne; base += 64) {
    const int k = base + lane;
    int nf = 0;
    if (k < nligne && t.flag[k] == F_UNKNOWN) {
      const li64 *r = t.val + t.ref[k] * t.W;
      int ff = F_ZERO;
      for (int j = nvar + 1; j < ncol; j++) {
        const int fj = sgn_flag(r[j]);
        if (fj != F_ZERO && fj != ff) {
          if (ff == F_ZERO)
            ff = fj;
          else {
            ff = F_UNKNOWN;
            break;
          }
        }
      }
      const int fc = sgn_flag(r[nvar]);  // constant term, traiter.c:138-140
      if (ff == F_PLUS) {
        if (fc == F_MINUS) ff = F_UNKNOWN;
      } else if (ff == F_ZERO) {
        ff = fc;
      } else if (ff == F_MINUS) {
        if (fc != F_MINUS) ff = F_UNKNOWN;
      }
      nf = ff;
    }
    const u64 neg = __ballot(nf == F_MINUS);
    const int stop = neg ? first64(neg) : 64;
    if (nf && lane <= stop) t.flag[k] = nf;
    if (neg) {
      wsync();
      return base + stop;
    }
  }
  wsync();
  return nligne;
}

// traiter.c:556-623 tab_sort_rows: selection sort of the real rows nvar..nligne-1 by
// max |trunc(coefficient / denominator)| over the unknowns (float keys, the reference's types)
static __device__ __forceinline__ int trunc_x86(double t) {
  return (!(t > -2147483649.0 && t < 2147483648.0)) ? (int)0x80000000 : (int)t;
}
static __device__ __forceinline__ int rdlane(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
// tab_sort_rows for 65 ... 128 rows to sort (round 4): the same selection sort -- the first row at or after i with the
// smallest key strictly below the maximum is swapped into place i, traiter.c:591-614 -- with two rows per lane and the rows'
// data left in LDS: keys in `skey` (128 ints), the swaps done there by lane 0.  No Compute_dual at this size.
static __device__ __noinline__ int sort_rows_tall(Tab t, int nvar, int nligne, int lane, LDS int *skey) {
  const int n = nligne - nvar;
  if (n > 128) return Q_WHY_ROWS | 256;
  u64 realm[2], below[2];
  int sv[2];
  bool rl[2];
  int smx = 0;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int r0 = lane + 64 * h, k = nvar + r0;
    int s = 0;
    bool real = false;
    if (r0 < n) {
      const int fl = t.flag[k], rf = t.ref[k];
      const i64 dn = t.den[k];
      real = !(fl & F_UNIT);
      if (real) {
        const li64 *r = t.val + rf * t.W;
        if (dn == 1) {
          for (int j = 0; j < nvar; j++) {
            const i64 v = r[j];
            const int q = v == (i64)(int)v ? (int)v : (int)0x80000000;
            const int a = q < 0 ? (int)(0u - (unsigned)q) : q;
            s = s > a ? s : a;
          }
        } else {
          const double d = qdouble(dn);
          for (int j = 0; j < nvar; j++) {
            const int q = trunc_x86(qdouble(r[j]) / d);
            const int a = q < 0 ? (int)(0u - (unsigned)q) : q;
            s = s > a ? s : a;
          }
        }
      }
    }
    sv[h] = s;
    rl[h] = real;
    realm[h] = __ballot(real);
    const int m = wave_max_i(real ? s : 0);
    smx = m > smx ? m : smx;
    if (r0 < n) skey[r0] = __float_as_int((float)(double)s);  // non-negative floats order like their bit patterns
  }
  const double smax = (double)smx;
#pragma unroll
  for (int h = 0; h < 2; h++) below[h] = __ballot(rl[h] && (double)(float)(double)sv[h] < smax);
  wsync();
  if (!(below[0] | below[1])) return 0;  // no key is below the maximum: no row moves
  for (int i = 0; i < n; i++) {
    const int hi = i >> 6, bi = i & 63;
    if (!((realm[hi] >> bi) & 1)) continue;
    const u64 c0 = hi == 0 ? below[0] & ~((1ull << bi) - 1) : 0ull;
    const u64 c1 = hi == 0 ? below[1] : below[1] & ~((1ull << bi) - 1);
    if (!(c0 | c1)) break;
    const int k0 = ((c0 >> lane) & 1) ? skey[lane] : 0x7fffffff, k1 = ((c1 >> lane) & 1) ? skey[64 + lane] : 0x7fffffff;
    const int best = wave_min_i(k0 < k1 ? k0 : k1);
    const u64 m0 = __ballot(k0 == best && ((c0 >> lane) & 1)), m1 = __ballot(k1 == best && ((c1 >> lane) & 1));
    const int p = m0 ? first64(m0) : 64 + first64(m1);
    if (p == i) continue;
    if (lane == 0) {  // rows i and p trade places (traiter.c:604-612)
      const int ki = nvar + i, kp = nvar + p;
      const int f = t.flag[ki], r = t.ref[ki], q = skey[i];
      const i64 d = t.den[ki];
      t.flag[ki] = t.flag[kp];
      t.ref[ki] = t.ref[kp];
      t.den[ki] = t.den[kp];
      skey[i] = skey[p];
      t.flag[kp] = f;
      t.ref[kp] = r;
      t.den[kp] = d;
      skey[p] = q;
    }
    {
      const u64 bit_i = (below[hi] >> bi) & 1;
      const int hp = p >> 6, bp = p & 63;
      below[hp] = (below[hp] & ~(1ull << bp)) | (bit_i << bp);
      below[hi] |= 1ull << bi;
    }
    wsync();
  }
  return 0;
}

// pos != null (Compute_dual): pos[i] = the logical row inequality i (row nvar + i before the sort) ends up in; unit
// rows among nvar.. count as inequality 0, later rows overwriting earlier ones -- the reference never sets their
// `ineq` (traiter.c:577-578 vs 617-618), zero-filled as the oracle and the reference's own fixtures have it.
static __device__ __noinline__ int sort_rows(Tab t, int nvar, int nligne, int lane, LDS unsigned short *pos, LDS int *skey) {
  const int n = nligne - nvar;  // rows to sort: at most 64 (a lane each) -- or up to 128, two per lane (sort_rows_tall)
  if (n > 64) return (pos || !skey) ? (Q_WHY_ROWS | 256) : sort_rows_tall(t, nvar, nligne, lane, skey);
  // lane l holds logical row nvar + l (flag, slot, denominator, key); the selection sort swaps lanes
  const int k = nvar + lane;
  int fl = 0, rf = 0, s = 0;
  i64 dn = 1;
  bool real = false;
  if (lane < n) {
    fl = t.flag[k];
    rf = t.ref[k];
    dn = t.den[k];
    real = !(fl & F_UNIT);
  }
  if (real) {
    const li64 *r = t.val + rf * t.W;
    if (dn == 1) {  // x / 1.0 is x, and (int)x is x itself when it fits an int (every row of a fresh tableau)
      for (int j = 0; j < nvar; j++) {
        const i64 v = r[j];
        const int q = v == (i64)(int)v ? (int)v : (int)0x80000000;
        const int a = q < 0 ? (int)(0u - (unsigned)q) : q;
        s = s > a ? s : a;
      }
    } else {
      const double d = qdouble(dn);
      for (int j = 0; j < nvar; j++) {
        const int q = trunc_x86(qdouble(r[j]) / d);
        const int a = q < 0 ? (int)(0u - (unsigned)q) : q;  // abs() incl. INT_MIN
        s = s > a ? s : a;  // (double)INT_MIN never wins against s >= 0
      }
    }
  }
  int oi = real ? lane : 0;  // the inequality this lane's row is (Compute_dual)
  const u64 realm = __ballot(real);
  int smx = 0;
  for (u64 c = realm; c; c &= c - 1) {
    const int sj = rdlane(s, first64(c));
    smx = sj > smx ? sj : smx;
  }
  const double smax = (double)smx;
  int kb = __float_as_int((float)(double)s);  // non-negative floats order like their bit patterns
  u64 below = __ballot(real && (double)(float)(double)s < smax);
  bool moved = false;
  if (below)  // (else no key is below the maximum: no row moves)
  for (int i = 0; i < n; i++) {
    if (!((realm >> i) & 1)) continue;
    u64 c = below & ~((1ull << i) - 1);
    if (!c) break;
    int best = 0x7fffffff, p = -1;
    for (; c; c &= c - 1) {  // the first row at or after i with the smallest key below the maximum
      const int j = first64(c), kj = rdlane(kb, j);
      if (kj < best) {
        best = kj;
        p = j;
      }
    }
    if (p == i) continue;
    {  // rows i and p trade places (traiter.c:604-612)
      const int f_i = rdlane(fl, i), f_p = rdlane(fl, p), r_i = rdlane(rf, i), r_p = rdlane(rf, p);
      const int k_i = rdlane(kb, i), k_p = rdlane(kb, p), o_i = rdlane(oi, i), o_p = rdlane(oi, p);
      const i64 dn_i = qrdlane(dn, i), dn_p = qrdlane(dn, p);
      if (lane == i) {
        fl = f_p;
        rf = r_p;
        kb = k_p;
        oi = o_p;
        dn = dn_p;
      } else if (lane == p) {
        fl = f_i;
        rf = r_i;
        kb = k_i;
        oi = o_i;
        dn = dn_i;
      }
      const u64 bi = (below >> i) & 1;
      below = (below & ~(1ull << p) & ~(1ull << i)) | (1ull << i) | (bi << p);
      moved = true;
    }
  }
  if (moved) {
    if (lane < n) {
      t.flag[k] = fl;
      t.ref[k] = rf;
      t.den[k] = dn;
    }
    wsync();
  }
  if (pos) {
    if (lane < 64) pos[lane] = 0;
    wsync();
    const u64 zero = __ballot(lane < n && oi == 0);  // the row that was inequality 0, and every unit row
    if (lane < n && oi > 0) pos[oi] = (unsigned short)k;
    if (zero && lane == 63 - __builtin_clzll(zero)) pos[0] = (unsigned short)k;  // the last of them wins
    wsync();
  }
  return 0;
}

// traiter.c:345-548 pivoter (with choisir_piv, traiter.c:297-341, as a tournament over the rows);
// returns -1 when the pivot row has no positive entry among the unknowns
static __device__ __noinline__ int pivot_step(Tab t, int pivi, int nvar, int ncol, int nligne, int lane) {
  const int W = t.W;
  int bad = 0;
  const int pslot = t.ref[pivi];
  const i64 p = lane < ncol ? t.val[pslot * W + lane] : 0;
  u64 tied = __ballot(lane < nvar && p > 0);
  if (!tied) return -1;
  for (int k = 0; k < nligne && popc64(tied) > 1; k++) {
    const int fl = t.flag[k], rf = t.ref[k];
    if (fl & F_UNIT) {  // its own column has the only positive ratio there: every other tied column is smaller
      tied &= ~(1ull << rf);
      continue;
    }
    const i64 v = lane < ncol ? t.val[rf * W + lane] : 0;
    const bool in = (tied >> lane) & 1;
    if (!__ballot(in && v != 0)) continue;
    int c = first64(tied);
    // each round moves to a strictly smaller ratio, so at most 64 rounds on true integers; products that
    // overflowed compare as garbage and could go round in circles: the problem is handed back then
    for (int round = 0;; round++) {
      const i64 pc = bcast(p, c), vc = bcast(v, c);
      const i64 x = csub(cmul(pc, v, bad), cmul(vc, p, bad), bad);
      const u64 less = __ballot(in && x < 0);
      if (!less) {
        tied = __ballot(in && x == 0);
        break;
      }
      if (round >= 64 || __any(bad)) {
        bad |= Q_WHY_OVERFLOW;
        tied = 1ull << c;
        break;
      }
      c = first64(less);
    }
    if (!tied) {  // (only with garbage from an overflow)
      bad |= Q_WHY_OVERFLOW;
      tied = 1ull << c;
    }
  }
  if (__any(bad)) return bad;
  const int pivj = first64(tied);
  const i64 pivot = bcast(p, pivj), dpiv = t.den[pivi];
  // the determinant in limbs, traiter.c:412-446 (uniform values; lane 0 publishes them).  A pivot of 1
  // over a denominator of 1 multiplies a limb that still has room by 1: nothing to do.
  constexpr int EBW = 8 * (int)sizeof(i64);  // bits of an Entier (traiter.c:430: lllog2(limb) + lllog2(pivot) < 8 * sizeof(Entier))
  if (pivot != 1 || dpiv != 1 || blen(t.det[0]) + 1 >= EBW) {
    i64 d = gcd64(pivot, dpiv);
    const i64 ppivot = d == 1 ? pivot : quo(pivot, d);
    i64 dppiv = d == 1 ? dpiv : quo(dpiv, d);
    int ldet = *t.ldet;
    i64 dt[MAXDET];
#pragma unroll
    for (int i = 0; i < MAXDET; i++) dt[i] = t.det[i];
#pragma unroll
    for (int i = 0; i < MAXDET; i++)
      if (i < ldet && dppiv != 1) {
        d = gcd64(dt[i], dppiv);
        if (d != 1) {
          dt[i] = quo(dt[i], d);
          dppiv = quo(dppiv, d);
        }
      }
    if (dppiv != 1) bad |= Q_WHY_OVERFLOW;  // "Integer overflow", traiter.c:424
    bool placed = false;
#pragma unroll
    for (int i = 0; i < MAXDET; i++)
      if (!placed && i < ldet && blen(dt[i]) + blen(ppivot) < EBW) {
        dt[i] *= ppivot;
        placed = true;
      }
    if (!placed) {
      if (ldet + 1 >= MAXDET) {
        bad |= Q_WHY_OVERFLOW;  // traiter.c:442
      } else {
#pragma unroll
        for (int i = 0; i < MAXDET; i++)
          if (i == ldet) dt[i] = ppivot;
        ldet++;
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < MAXDET; i++) t.det[i] = dt[i];
      *t.ldet = ldet;
    }
    wsync();
    if (__any(bad)) return bad;
  }
  // eliminate column pivj from every other real row, traiter.c:467-502: lane k rewrites row k, every
  // row at once, walking the columns (the pivot row is read from LDS, the same address for every lane).
  // The same pass finds the unit row of column pivj and refreshes the sign hints (traiter.c:518-529):
  // the row's new entry in column pivj is dpiv * foo / d, which has the sign of foo.
  int ku = -1;
  for (int base = 0; base < nligne; base += 64) {
    const int k = base + lane;
    const bool in = k < nligne;
    const int fl = in ? t.flag[k] : F_UNIT, rfk = in ? t.ref[k] : -1;
    const u64 mu = __ballot(in && (fl & F_UNIT) && rfk == pivj);
    if (mu) ku = base + first64(mu);
    bool act = false;
    i64 lpiv = 1, fo = 0, g = 1;
    li64 *r = t.val;
    if (in && k != pivi && !(fl & F_UNIT)) {
      r = t.val + rfk * W;
      const i64 foo = r[pivj], oden = t.den[k];
      if (foo != 0 || oden != 1) {  // else: multipliers (1, 0) and g = 1, the row keeps its bits
        act = true;
        const i64 d = gcd64(pivot, foo);
        lpiv = d == 1 ? pivot : quo(pivot, d);
        fo = d == 1 ? foo : quo(foo, d);
        g = oden == 1 ? lpiv : cmul(lpiv, oden, bad);
      }
      const int fff = sgn_flag(foo);
      if (fff != F_ZERO && fff != fl) t.flag[k] = fl == F_ZERO ? (fff == F_MINUS ? F_UNKNOWN : fff) : F_UNKNOWN;
    }
    if (__ballot(act)) {
      const li64 *prow = t.val + pslot * W;
      // operands below 2^31: a product is below 2^62 and the difference of two fits, no check needed
      const bool small = fits32(lpiv) && fits32(fo) && fits32(dpiv);
#pragma unroll 4
      for (int j = 0; j < ncol; j++) {
        const i64 pj = prow[j];
        if (act) {
          const i64 v = r[j];
          i64 z;
          if (small && fits32(v) && fits32(pj))
            z = j == pivj ? mul32(dpiv, fo) : mul32(v, lpiv) - mul32(pj, fo);
          else
            z = j == pivj ? cmul(dpiv, fo, bad) : csub(cmul(v, lpiv, bad), cmul(pj, fo, bad), bad);
          r[j] = z;
        }
      }
      // gcd of g and the whole row (integrer.c:43-50 folds it the same way, entry by entry), four entries
      // at a time: usually g divides them all, or the gcd drops to 1 at once
      uE G = uabs(g);
      for (int j = 0; j < ncol; j += 4) {
        if (!__ballot(act && G != 1)) break;
        if (act && G != 1) {
          uE m[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const i64 z = j + q < ncol ? r[j + q] : 0;
            m[q] = z ? umod(uabs(z), G) : 0;
          }
          if (m[0] | m[1] | m[2] | m[3]) {
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (m[q] && G != 1) G = (uE)gcd64((i64)G, (i64)umod(m[q], G));
          }
        }
      }
      if (__ballot(act && G != 1)) {
        // exact division by G = 2^tz * odd: shift, then multiply by the inverse of the odd part modulo 2^64
        const int tz = qctz(G);
        const uE od = G >> tz;
        const uE inv = qinv(od);
        const bool dv = act && G != 1;
#pragma unroll 4
        for (int j = 0; j < ncol; j++)
          if (dv) r[j] = (i64)((uE)(r[j] >> tz) * inv);
        if (dv) g = (i64)((uE)(g >> tz) * inv);
      }
      if (act) t.den[k] = g;
    }
  }
  if (ku < 0) {
    return bad | Q_WHY_OTHER;
  }
  // swap roles, traiter.c:503-516: the unit row of pivj becomes real (in the pivot row's slot)
  if (lane < ncol) t.val[pslot * W + lane] = lane == pivj ? dpiv : cneg(p, bad);
  if (lane == 0) {
    t.flag[ku] = F_PLUS;
    t.ref[ku] = pslot;
    t.den[ku] = pivot;
    t.flag[pivi] = F_UNIT | F_ZERO;
    t.den[pivi] = 1;
    t.ref[pivi] = pivj;
  }
  wsync();
  return bad;
}

// integrer.c:98-150 bezout
static __device__ __forceinline__ i64 bezout(i64 x, i64 y, i64 delta, int &bad) {
  i64 a = 1, b = 0, c = 0, d = 1, u = y, v = delta;
  for (int guard = 0; guard < 200; guard++) {
    const i64 q = floordiv(u, v, bad), r = pmod(u, v);
    if (r == 0) break;
    u = v;
    v = r;
    const i64 e = csub(a, cmul(q, c, bad), bad), f = csub(b, cmul(q, d, bad), bad);
    a = c;
    b = d;
    c = e;
    d = f;
  }
  if (v != 1) return 0;
  return pmod(cmul(c, x, bad), delta);
}

// the cut of row i (integrer.c:342-400), one column per lane
struct Cut {
  i64 c;
  bool ok_var, ok_const, ok_parm;
};
static __device__ __forceinline__ Cut make_cut(const Tab &t, int i, int nvar, int ncol, int bigparm, int lane) {
  Cut q;
  const i64 D = t.den[i];
  const i64 v = lane < ncol ? t.val[t.ref[i] * t.W + lane] : 0;
  i64 c = 0;
  if (lane < nvar)
    c = pmod(v, D);
  else if (lane < ncol && lane != bigparm)  // the big parameter is a multiple of everything
    c = -pmod(-v, D);
  q.c = c;
  q.ok_var = __ballot(lane < nvar && c > 0) != 0;
  q.ok_const = bcast(c, nvar) != 0;
  q.ok_parm = __ballot(lane > nvar && lane < ncol && c != 0) != 0;
  return q;
}

// deepest cut, integrer.c:417-438 (constant cuts only)
static __device__ __forceinline__ i64 deepen(i64 c, i64 D, int nvar, int lane, int &bad) {
  const i64 cst = bcast(c, nvar);
  i64 tt = -cst;
  const i64 delta = gcd64(tt, D), tau = quo(tt, delta), dd = quo(D, delta);
  tt = dd - 1;
  i64 lambda = bezout(tt, tau, dd, bad);
  tt = gcd64(lambda, D);
  for (int guard = 0; tt != 1 && guard < 100000; guard++) {
    lambda = cadd(lambda, dd, bad);
    tt = gcd64(lambda, D);
  }
  if (tt != 1) bad |= Q_WHY_OTHER;
  if (lane < nvar) return pmod(cmul(lambda, c, bad), D);
  if (lane == nvar) return -(D - pmod(cmul(c, lambda, bad), D));
  return c;
}

// append a cut as logical row nligne in slot ni (flag Minus, denominator D); false: no room
static __device__ __forceinline__ bool append_row(Tab &t, int nligne, int ni, i64 c, i64 D, int lane) {
  if (nligne >= t.rows_cap || ni >= t.slots_cap) return false;
  if (lane < t.W) t.val[ni * t.W + lane] = c;
  if (lane == 0) {
    t.flag[nligne] = F_MINUS;
    t.ref[nligne] = ni;
    t.den[nligne] = D;
  }
  wsync();
  return true;
}

// traiter() of a tableau without parameters, integer solve (the sub-problems of compa_test and the
// context test): true when the first cell of its tape would not be Nil
// Result word: bit 0 = a solution exists, bits 1..15 = reason bits (per lane), bits 16.. = pivots.
// `budget`: pivots the problem may still spend (Q_PIVOT_BUDGET less what it has used): beyond it the problem is handed back.
static __device__ __noinline__ int solve_plain(Tab t, int nvar, int ni, int lane, int deepest, int budget) {
  const int ncol = nvar + 1;
  int bad = sort_rows(t, nvar, nvar + ni, lane, nullptr, nullptr), pivots = 0, found = 0;
  for (int guard = 0; guard < 30000 && !__any(bad); guard++) {  // (the pivot count has 15 bits of the result word)
    const int nligne = nvar + ni;
    int pivi = first_flagged(t, F_MINUS, nligne, lane);
    if (pivi >= nligne) pivi = classify_rows(t, nvar, ncol, -1, nligne, lane);
    if (pivi >= nligne) {
      // integrer.c:305-534 with constant cuts only
      int i;
      bool nil = false;
      const u64 frac = __ballot(lane < nvar && !(t.flag[lane] & F_UNIT) && t.den[lane] != 1);  // rows that may be fractional
      for (i = frac ? first64(frac) : nvar; i < nvar; i++) {
        if (!((frac >> i) & 1)) continue;
        Cut q = make_cut(t, i, nvar, ncol, -1, lane);
        if (!q.ok_const) continue;  // integral row
        if (!q.ok_var) {            // constant fractional, nothing to cut with
          nil = true;
          break;
        }
        const i64 D = t.den[i];
        i64 c = q.c;
        if (deepest) c = deepen(c, D, nvar, lane, bad);
        if (lane >= ncol) c = 0;
        if (!append_row(t, nligne, ni, c, D, lane)) bad |= Q_WHY_ROWS | 512;
        pivi = nligne;
        ni++;
        break;
      }
      if (nil || __any(bad)) break;
      if (i >= nvar) {  // every unknown integral: a solution
        found = 1;
        break;
      }
    }
    pivots++;
    const int pr = pivot_step(t, pivi, nvar, ncol, nvar + ni, lane);
    if (__any(pr < 0)) break;  // no positive entry in the pivot row: Nil
    bad |= pr;
    if (guard == 29999 || pivots > budget) bad |= Q_WHY_OTHER;  // (the pivot budget: see the kernel)
  }
  return found | (bad << 1) | (pivots << 16);
}

// the tableau "context (+ one more row)" of compa_test / the context test (traiter.c:196-233,
// maind.c:196-203): nparm unit rows, the nc context rows, `extra` as the last row
static __device__ __forceinline__ int build_sub(Tab &s, const li64 *ctx, int CW, int nparm, int nc, bool has_extra, i64 extra,
                                         int lane) {
  const int ni = nc + (has_extra ? 1 : 0);
  if (nparm + ni > s.rows_cap || ni > s.slots_cap) return -1;
  for (int base = 0; base < nparm + ni; base += 64) {
    const int k = base + lane;
    if (k < nparm) {
      s.flag[k] = F_UNIT;
      s.ref[k] = k;
      s.den[k] = 1;
    } else if (k < nparm + ni) {
      s.flag[k] = F_UNKNOWN;
      s.ref[k] = k - nparm;
      s.den[k] = 1;
    }
  }
  for (int r = 0; r < nc; r++)
    if (lane < s.W) s.val[r * s.W + lane] = lane <= nparm ? ctx[r * CW + lane] : 0;
  if (has_extra && lane < s.W) s.val[nc * s.W + lane] = lane <= nparm ? extra : 0;
  if (lane == 0) {
    *s.ldet = 1;
    s.det[0] = 1;
  }
  wsync();
  return ni;
}

struct Tape {
  i64 *cell;  // global: 3 words per cell (kind, param1, param2)
  int n, cap;
};
static __device__ __forceinline__ void tape_put(Tape &tp, int at, int kind, i64 a, i64 b) {
  if (at < tp.cap) {
    tp.cell[3 * (size_t)at] = kind;
    tp.cell[3 * (size_t)at + 1] = a;
    tp.cell[3 * (size_t)at + 2] = b;
  }
}

// Which parameter, if any, is already the quotient a parametric cut needs (find_parm, integrer.c:258-291)?
// Parameter p is floor(-(c . (1, params)) / D) when the context holds the two rows that defined it
// (integrer.c:156-227): +(c_params | D at p | c0 + D - 1) and -(c_params | D at p | c0), nothing right of p.
// cutv = c0 | c_params | D in LDS; a lane per context row; the highest such p, -1 if none.
static __device__ __forceinline__ int find_quotient(const li64 *ctx, int CW, int nc, int nparm, const li64 *cutv, int lane,
                                             int &bad) {
  if (cutv[nparm] != 0) return -1;  // the last parameter takes part in the cut: it cannot be the quotient's
  const i64 c0 = cutv[0], D = cutv[1 + nparm];
  const i64 cplus = csub(cadd(c0, D, bad), 1, bad), cminus = cneg(c0, bad), Dm = cneg(D, bad);
  for (int p = nparm - 1; p >= 0; --p) {
    if (cutv[1 + p] != 0) break;
    u64 has_plus = 0, has_minus = 0;
    for (int base = 0; base < nc; base += 64) {
      const int k = base + lane;
      bool mp = k < nc, mm = mp;
      if (mp) {
        const li64 *v = ctx + k * CW;
        for (int col = 0; col < p; col++) {
          const i64 a = v[col], c = cutv[1 + col];
          mp = mp && a == c;
          mm = mm && a == cneg(c, bad);
        }
        bool tail = true;
        for (int col = p + 1; col < nparm; col++) tail = tail && v[col] == 0;
        mp = mp && tail && v[p] == D && v[nparm] == cplus;
        mm = mm && tail && v[p] == Dm && v[nparm] == cminus;
      }
      has_plus |= __ballot(mp);
      has_minus |= __ballot(mm);
    }
    if (has_plus && has_minus) return p;
  }
  return -1;
}

static __device__ __forceinline__ void run(const QProb *probs, const w64 *input, i64 *stack, i64 *cells, int *out, int nprob,
                                    const QCaps &cap) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr size_t EB = sizeof(i64);  // bytes of an entry
  const int pi = blockIdx.x;
  if (pi >= nprob) return;
  const QProb P = probs[pi];
  const long long t_start = wall_clock64();
  // No problem keeps its wave for more than Q_PIVOT_BUDGET pivots (its own and those of its compa_test sub-problems;
  // about two seconds of one wave): whatever is still running then is handed back to the host schedulers.  Every loop
  // below is bounded on its own; this bounds their product -- by a count, so that which side serves a problem does not
  // depend on the load of the GPU.  The clock (wall_clock64 ticks at 100 MHz) stays as a last resort at 60 seconds.
  constexpr int Q_PIVOT_BUDGET = 250000;
  const long long deadline = t_start + 6000000000ll;
  Wv w;
  w.lane = threadIdx.x;
  w.bad = 0;
  w.pivots = 0;
  w.deepest = cap.deepest;
  const int lane = w.lane;

  // ---- LDS carve-up: [main: den | val | ctx | flag | ref | state] [sub: den | val | det | flag | ref] [cutv]
  LDS unsigned char *q = (LDS unsigned char *)smem;
  LDS unsigned char *const q0 = q;
  Tab M, S;
  M.den = (li64 *)q;
  q += EB * (size_t)cap.R;
  M.val = (li64 *)q;
  q += EB * (size_t)cap.S * cap.W;
  li64 *ctx = (li64 *)q;
  q += EB * (size_t)cap.CR * cap.CW;
  M.flag = (lint *)q;
  q += 4 * (size_t)cap.R;
  M.ref = (lint *)q;
  q += 4 * (size_t)cap.R;
  LDS QState *st = (LDS QState *)q;
  q += sizeof(QState);
  const size_t main_words = (size_t)(q - q0) / EB;  // (entries: R and SR are even, the image is whole 16-byte units)
  M.ldet = &st->ldet;
  M.det = st->det;
  M.W = cap.W;
  M.rows_cap = cap.R;
  M.slots_cap = cap.S;
  S.den = (li64 *)q;
  q += EB * (size_t)cap.SR;
  S.val = (li64 *)q;
  q += EB * (size_t)cap.SS * cap.CW;
  S.det = (li64 *)q;
  q += EB * MAXDET;
  S.flag = (lint *)q;
  q += 4 * (size_t)cap.SR;
  S.ref = (lint *)q;
  q += 4 * (size_t)cap.SR;
  S.ldet = (lint *)q;
  q += 16;
  S.W = cap.CW;
  S.rows_cap = cap.SR;
  S.slots_cap = cap.SS;
  li64 *cutv = (li64 *)q;  // [CW + 2]
  q += EB * ((size_t)cap.CW + 2);
  LDS int *skey = (LDS int *)q;  // [128] sort keys of a tall tableau (sort_rows_tall)

  i64 *my_stack = stack + (size_t)pi * cap.depth * main_words;
  Tape tape;
  tape.cell = cells + (size_t)pi * cap.cells * 3;
  tape.n = 0;
  tape.cap = cap.cells;
  int sp = 0;

  int nvar = P.nvar, nparm = P.nparm, ni = P.ni, nc = P.nc;
  const int bigparm = P.bigparm, CW = cap.CW, W = cap.W;
  const bool integer = P.nq != 0;
  const bool dual = (P.flags & Q_DUAL) != 0 && !integer;  // the dual goes with rational solves only (piplib.c:854-857)
  int result = Q_DONE;

  // ---- load: zero the main image, rows Unknown with denominator 1 under nvar unit rows (tab.c:158-248)
  for (size_t k = lane; k < main_words; k += 64) ((li64 *)q0)[k] = 0;
  wsync();
  {
    const int ncol = nvar + nparm + 1;
    const w64 *in = input + P.in_off;
    for (int r = 0; r < ni; r++)
      if (lane < ncol) M.val[r * W + lane] = in[(size_t)r * ncol + lane];
    const w64 *cin = in + (size_t)ni * ncol;
    for (int r = 0; r < nc; r++)
      if (lane <= nparm) ctx[r * CW + lane] = cin[(size_t)r * (nparm + 1) + lane];
    for (int base = 0; base < nvar + ni; base += 64) {
      const int k = base + lane;
      if (k < nvar) {
        M.flag[k] = F_UNIT;
        M.ref[k] = k;
        M.den[k] = 1;
      } else if (k < nvar + ni) {
        M.flag[k] = F_UNKNOWN;
        M.ref[k] = k - nvar;
        M.den[k] = 1;
      }
    }
    if (lane == 0) {
      st->ldet = 1;
      st->det[0] = 1;
    }
  }
  wsync();
  if (cap.simplify && P.nq) {  // tab_simplify (tab.c:396-427, maind.c:190-196), a row per lane
    for (int pass = 0; pass < 2; pass++) {
      li64 *rows = pass ? ctx : M.val;
      const int nrows = pass ? nc : ni, stride = pass ? CW : W, width = pass ? nparm + 1 : nvar + nparm + 1;
      const int cst = pass ? nparm : nvar;
      for (int base = 0; base < nrows; base += 64) {
        const int k = base + lane;
        if (k < nrows) {
          li64 *r = rows + k * stride;
          i64 g = 0;
          for (int j = 0; j < width; j++) {
            if (j == cst) continue;
            g = gcd64(g, r[j]);
            if (g == 1) break;
          }
          if (g != 0 && g != 1)
            for (int j = 0; j < width; j++) r[j] = j == cst ? floordiv(r[j], g, w.bad) : quo(r[j], g);
        }
      }
    }
    wsync();
  }

  // ---- maind.c:196-203 / piplib.c:813-823: is the context empty?
  if (nc && !(P.flags & Q_NO_CONTEXT_TEST)) {
    const int sni = build_sub(S, ctx, CW, nparm, nc, false, 0, lane);
    if (sni < 0)
      w.bad |= Q_WHY_ROWS | 1024;
    else {
      const int r = solve_plain(S, nparm, sni, lane, w.deepest, Q_PIVOT_BUDGET - w.pivots);
      w.bad |= (r >> 1) & 0x7fff;
      w.pivots += r >> 16;
      if (!(r & 1) && !BAD(w)) result = Q_VOID;
    }
  }

  // ---- traiter(), traiter.c:628-791, as a state machine: DECIDE (the head of the reference's loop) ->
  // PIVOT (its `pirouette` label) or LEAVE (the call returns: the caller's "else" state is popped)
  if (result == Q_DONE && !BAD(w)) {
    enum { DECIDE = 0, PIVOT = 1, LEAVE = 2 };
    int pivi = 0, next = DECIDE;
    bool enter = true, finished = false;
    for (int guard = 0; guard < 2000000 && !finished; guard++) {
      if (w.pivots > Q_PIVOT_BUDGET || wall_clock64() > deadline) w.bad |= Q_WHY_OTHER;
      if (BAD(w)) break;
      if (next == DECIDE) {
        if (enter) {
          w.bad |= sort_rows(M, nvar, nvar + ni, lane, dual ? st->pos : nullptr, skey);
          if (dual && lane == 0) st->ni0 = ni;
          wsync();
          enter = false;
          if (BAD(w)) break;
        }
        const int nligne = nvar + ni, ncol = nvar + nparm + 1;
        pivi = first_flagged(M, F_MINUS, nligne, lane);
        if (pivi >= nligne) pivi = classify_rows(M, nvar, ncol, bigparm, nligne, lane);
        if (pivi >= nligne && nparm > 0) {
          // compa_test, traiter.c:162-243
          if (nparm >= PIPAMD_MAXPARM) w.bad |= Q_WHY_OTHER;
          for (int i = first_flagged(M, F_CRITIC | F_UNKNOWN, nligne, lane); i < nligne && !BAD(w); i++) {
            const int fl = M.flag[i];
            if (!(fl & (F_CRITIC | F_UNKNOWN))) continue;
            const i64 v = lane < ncol ? M.val[M.ref[i] * W + lane] : 0;
            const bool critic = __ballot(lane < nvar && v > 0) == 0;
            // lane j <= nparm of the new context row: parameters, then the constant
            const i64 vc = bcast(v, nvar);
            const i64 vp = qshfl(v, (lane + nvar + 1) & 63);  // lane j < nparm: column nvar+1+j
            i64 ex = lane < nparm ? vp : (lane == nparm ? (critic ? vc : csub(vc, 1, w.bad)) : 0);
            int sni = build_sub(S, ctx, CW, nparm, nc, true, ex, lane);
            if (sni < 0) {
              w.bad |= Q_WHY_ROWS | 1024;
              break;
            }
            // "row >= 1" (>= 0 for a critical row), then "-row >= 1", over the context
            bool can[2] = {false, false};
            for (int sg = 0; sg < 2; sg++) {
              if (sg) {
                ex = lane < nparm ? cneg(vp, w.bad) : (lane == nparm ? csub(cneg(vc, w.bad), 1, w.bad) : 0);
                sni = build_sub(S, ctx, CW, nparm, nc, true, ex, lane);
              }
              const int r = solve_plain(S, nparm, sni, lane, w.deepest, Q_PIVOT_BUDGET - w.pivots);
              w.bad |= (r >> 1) & 0x7fff;
              w.pivots += r >> 16;
              can[sg] = r & 1;
            }
            const bool can_pos = can[0], can_neg = can[1];
            int nf;
            if (can_pos && can_neg)
              nf = critic ? F_CRITIC : F_UNKNOWN;
            else if (can_neg)
              nf = F_MINUS;
            else
              nf = can_pos ? F_PLUS : F_ZERO;
            if (lane == 0) M.flag[i] = nf;
            wsync();
            if (nf == F_MINUS) break;
          }
          if (BAD(w)) break;
          pivi = first_flagged(M, F_MINUS, nligne, lane);
        }
        if (pivi < nligne) {
          next = PIVOT;
        } else {
          pivi = first_flagged(M, F_CRITIC, nligne, lane);
          if (pivi >= nligne) pivi = first_flagged(M, F_UNKNOWN, nligne, lane);
          if (pivi < nligne) {
            // ---- the quast forks on the sign of row pivi, traiter.c:695-759
            if (nparm >= PIPAMD_MAXPARM || nc >= cap.CR || sp >= cap.depth || tape.n + nparm + 3 >= tape.cap) {
              w.bad |= sp >= cap.depth ? Q_WHY_STACK : (nc >= cap.CR ? (Q_WHY_ROWS | 4096) : (nparm >= PIPAMD_MAXPARM ? Q_WHY_OTHER : Q_WHY_TAPE));
              break;
            }
            const i64 v = lane < ncol ? M.val[M.ref[pivi] * W + lane] : 0;
            const i64 vc = bcast(v, nvar);
            const i64 vp = qshfl(v, (lane + nvar + 1) & 63);
            i64 g = 0;
            for (int j = 0; j < nparm; j++) g = gcd64(g, bcast(vp, j));
            if (!integer) g = gcd64(g, vc);
            if (g == 0) {
              w.bad |= Q_WHY_OTHER;
              break;
            }
            const i64 cr =
                lane < nparm ? quo(vp, g) : (lane == nparm ? (integer ? floordiv(vc, g, w.bad) : quo(vc, g)) : 0);
            if (lane == 0) {
              tape_put(tape, tape.n, C_IF, 0, 0);
              tape_put(tape, tape.n + 1, C_FORM, nparm + 1, 0);
            }
            if (lane <= nparm) tape_put(tape, tape.n + 2 + lane, C_VAL, cr, 1);
            tape.n += nparm + 3;
            // the "else" state waits on the stack: row pivi negative, the negated condition in the context
            if (lane < CW)
              ctx[nc * CW + lane] =
                  lane < nparm ? cneg(cr, w.bad) : (lane == nparm ? cneg(cadd(cr, 1, w.bad), w.bad) : 0);
            if (lane == 0) {
              M.flag[pivi] = F_MINUS;
              st->nvar = nvar;
              st->nparm = nparm;
              st->ni = ni;
              st->nc = nc + 1;
              st->pivi = pivi;
            }
            wsync();
            {
              i64 *dst = my_stack + (size_t)sp * main_words;
              for (size_t k = lane; k < main_words; k += 64) dst[k] = ((const li64 *)q0)[k];
              sp++;
            }
            wsync();
            // the "then" branch: row pivi positive, the condition in the context; a new traiter() call
            if (lane < CW) ctx[nc * CW + lane] = lane <= nparm ? cr : 0;
            if (lane == 0) M.flag[pivi] = F_PLUS;
            wsync();
            nc++;
            enter = true;
            continue;  // next == DECIDE
          }
          // ---- every sign settled: the solution, or a cut
          bool solution = !integer, nil = false;
          if (integer) {
            // integrer.c:305-534
            int i;
            const u64 frac = __ballot(lane < nvar && !(M.flag[lane] & F_UNIT) && M.den[lane] != 1);  // rows that may be fractional
            for (i = frac ? first64(frac) : nvar; i < nvar; i++) {
              if (!((frac >> i) & 1)) continue;
              Cut qc = make_cut(M, i, nvar, ncol, bigparm, lane);
              if (!qc.ok_parm && !qc.ok_const) continue;  // integral row
              const i64 D = M.den[i];
              i64 c = qc.c;
              if (!qc.ok_parm) {
                if (!qc.ok_var) {  // constant fractional, nothing to cut with
                  nil = true;
                  break;
                }
                if (w.deepest) c = deepen(c, D, nvar, lane, w.bad);
                if (lane >= ncol) c = 0;
                if (!append_row(M, nligne, ni, c, D, lane)) w.bad |= Q_WHY_ROWS | 2048;
                break;
              }
              // parametric cut, integrer.c:487-520; cutv = constant | parameters | divisor
              if (lane >= nvar && lane < ncol) cutv[lane - nvar] = c;
              if (lane == 0) cutv[1 + nparm] = D;
              wsync();
              int parm = find_quotient(ctx, CW, nc, nparm, cutv, lane, w.bad);
              if (parm == -1) {
                // integrer.c:156-227 add_parm: a new parameter q = floor(-(cut . (1,p)) / D)
                if (nparm + 2 > CW || nc + 2 > cap.CR || ncol + 1 > W || nparm + 1 >= PIPAMD_MAXPARM ||
                    tape.n + nparm + 5 >= tape.cap) {
                  w.bad |= tape.n + nparm + 5 >= tape.cap ? Q_WHY_TAPE : (nparm + 1 >= PIPAMD_MAXPARM ? Q_WHY_OTHER : (Q_WHY_ROWS | 8192));
                  break;
                }
                const i64 c0 = cutv[0];
                const i64 cp = lane < nparm ? cutv[1 + lane] : 0;
                if (lane == 0) {
                  tape_put(tape, tape.n, C_NEW, nparm, 0);
                  tape_put(tape, tape.n + 1, C_DIV, 0, 0);
                  tape_put(tape, tape.n + 2, C_FORM, nparm + 1, 0);
                  tape_put(tape, tape.n + 3 + nparm, C_VAL, cneg(c0, w.bad), 1);
                  tape_put(tape, tape.n + 4 + nparm, C_VAL, D, 1);
                }
                if (lane < nparm) tape_put(tape, tape.n + 3 + lane, C_VAL, cneg(cp, w.bad), 1);
                tape.n += nparm + 5;
                // the constant column moves one to the right in every context row
                for (int base = 0; base < nc; base += 64) {
                  const int k = base + lane;
                  if (k < nc) {
                    ctx[k * CW + nparm + 1] = ctx[k * CW + nparm];
                    ctx[k * CW + nparm] = 0;
                  }
                }
                if (lane < CW) {  // 0 <= -(cut . (1,p)) - D q  and  -(cut . (1,p)) - D q <= D - 1
                  i64 a = 0, b = 0;
                  if (lane < nparm) {
                    b = cp;
                    a = cneg(cp, w.bad);
                  } else if (lane == nparm) {
                    a = cneg(D, w.bad);
                    b = D;
                  } else if (lane == nparm + 1) {
                    a = cneg(c0, w.bad);
                    b = cadd(csub(c0, 1, w.bad), D, w.bad);
                  }
                  ctx[nc * CW + lane] = a;
                  ctx[(nc + 1) * CW + lane] = b;
                }
                wsync();
                parm = nparm;
                nparm++;
                nc += 2;
              }
              if (!qc.ok_var) {  // assert(ok_var), integrer.c:499
                w.bad |= Q_WHY_OTHER;
                break;
              }
              // the cut row: the first ncol columns of the cut, the divisor added in the quotient's column
              if (lane >= ncol) c = 0;
              if (lane == nvar + 1 + parm) c = cadd(c, D, w.bad);
              if (!append_row(M, nligne, ni, c, D, lane)) w.bad |= Q_WHY_ROWS | 2048;
              break;
            }
            if (BAD(w)) break;
            if (i >= nvar)
              solution = true;
            else if (!nil) {
              pivi = nligne;  // the fresh cut row is negative: pivot on it
              ni++;
              next = PIVOT;
            }
          }
          if (nil) {
            if (tape.n + 1 >= tape.cap) {
              w.bad |= Q_WHY_TAPE;
              break;
            }
            if (lane == 0) tape_put(tape, tape.n, C_NIL, 0, 0);
            tape.n++;
            next = LEAVE;
          } else if (solution) {
            // solution(), traiter.c:255-271
            const int nc1 = nvar + nparm + 1;
            const int need = 1 + nvar * (nparm + 2);
            if (tape.n + need >= tape.cap) {
              w.bad |= Q_WHY_TAPE;
              break;
            }
            if (lane == 0) tape_put(tape, tape.n, C_LIST, nvar, 0);
            for (int i = 0; i < nvar; i++) {
              const int at = tape.n + 1 + i * (nparm + 2);
              const i64 d = M.den[i];
              const i64 v = row_at(M, i, lane, nc1);
              if (lane == 0) tape_put(tape, at, C_FORM, nparm + 1, 0);
              if (lane > nvar && lane < nc1) tape_put(tape, at + (lane - nvar), C_VAL, v, d);
              if (lane == nvar) tape_put(tape, at + nparm + 1, C_VAL, v, d);
            }
            tape.n += need;
            if (dual) {
              // solution_dual, traiter.c:273-294: one value per inequality of this call
              const int ni0 = st->ni0;
              if (tape.n + 1 + 2 * ni0 >= tape.cap) {
                w.bad |= Q_WHY_TAPE;
                break;
              }
              if (lane == 0) tape_put(tape, tape.n, C_LIST, ni0, 0);
              if (lane < ni0) {
                const int k = st->pos[lane];
                i64 v = 0, d = 1;
                if (M.flag[k] & F_UNIT) {  // valeur(tp, 0, unit column of row k) over Denom(tp, 0)
                  const int u = M.ref[k];
                  d = M.den[0];
                  v = (M.flag[0] & F_UNIT) ? (M.ref[0] == u ? d : 0) : M.val[M.ref[0] * W + u];
                }
                tape_put(tape, tape.n + 1 + 2 * lane, C_FORM, 1, 0);
                tape_put(tape, tape.n + 2 + 2 * lane, C_VAL, v, d);
              }
              tape.n += 1 + 2 * ni0;
            }
            next = LEAVE;
          }
        }
      }
      if (next == PIVOT) {
        next = DECIDE;
        w.pivots++;
        const int pr = pivot_step(M, pivi, nvar, nvar + nparm + 1, nvar + ni, lane);
        if (!__any(pr < 0)) w.bad |= pr;
        if (__any(pr < 0)) {
          if (tape.n + 1 >= tape.cap) {
            w.bad |= Q_WHY_TAPE;
            break;
          }
          if (lane == 0) tape_put(tape, tape.n, C_NIL, 0, 0);
          tape.n++;
          next = LEAVE;
        }
      }
      if (next == LEAVE) {
        if (sp == 0) {
          finished = true;
          break;
        }
        sp--;
        const i64 *src = my_stack + (size_t)sp * main_words;
        wsync();
        for (size_t k = lane; k < main_words; k += 64) ((li64 *)q0)[k] = src[k];
        wsync();
        nvar = st->nvar;
        nparm = st->nparm;
        ni = st->ni;
        nc = st->nc;
        pivi = st->pivi;
        next = PIVOT;  // traiter.c:758: the caller goes on with `pirouette` on the row now negative
      }
    }
    if (!finished) w.bad |= Q_WHY_OTHER;
  }
  int why = w.bad;
  for (int o = 32; o; o >>= 1) why |= __shfl_xor(why, o);
  if (why) result = Q_FALLBACK;
  if (lane == 0) {
    out[Q_OUT * pi] = result;
    out[Q_OUT * pi + 1] = result == Q_DONE ? tape.n : 0;
    out[Q_OUT * pi + 2] = w.pivots;
    out[Q_OUT * pi + 3] = why;
    out[Q_OUT * pi + 4] = (int)(wall_clock64() - t_start);  // 10 ns units (100 MHz), diagnostics
    out[Q_OUT * pi + 5] = tape.n;
  }
}

};  // struct QK

template <class QI>
__global__ __launch_bounds__(64) void pip_quast_kernel(const QProb *probs, const w64 *input, QI *stack, QI *cells, int *out,
                                                        int nprob, QCaps cap) {
  QK<QI>::run(probs, input, stack, cells, out, nprob, cap);
}

// cells of every finished problem, packed back to back: off[i] .. off[i+1] (a cell is `cw` raw words: three entries)
__global__ void pip_quast_pack_kernel(const w64 *cells, const w64 *off, w64 *packed, int cells_cap, int cw) {
  const int pi = blockIdx.x;
  const w64 lo = off[pi], n = off[pi + 1] - lo;
  const w64 *src = cells + (size_t)pi * cells_cap * cw;
  for (w64 k = threadIdx.x; k < cw * n; k += blockDim.x) packed[cw * lo + k] = src[k];
}

}  // namespace

// LDS image and stack frame of a launch, by entry width (ebits 64 or 128): EB bytes an entry
static size_t quast_state_bytes(int ebits) { return ebits == 128 ? sizeof(QK<w128>::QState) : sizeof(QK<w64>::QState); }
extern "C" size_t pipk_quast_lds_bytes(const QCaps *c, int ebits) {
  const size_t EB = ebits == 128 ? 16 : 8;
  size_t b = EB * (size_t)c->R + EB * (size_t)c->S * c->W + EB * (size_t)c->CR * c->CW + 8 * (size_t)c->R + quast_state_bytes(ebits);
  b += EB * (size_t)c->SR + EB * (size_t)c->SS * c->CW + EB * MAXDET + 8 * (size_t)c->SR + 16;
  b += EB * ((size_t)c->CW + 2) + 4 * 128;
  return (b + 15) & ~(size_t)15;
}
// entries (not bytes) of one frame of the fork stack
extern "C" size_t pipk_quast_frame_words(const QCaps *c, int ebits) {
  const size_t EB = ebits == 128 ? 16 : 8;
  return (EB * (size_t)c->R + EB * (size_t)c->S * c->W + EB * (size_t)c->CR * c->CW + 8 * (size_t)c->R + quast_state_bytes(ebits)) / EB;
}

static int g_quast_lds[2][64];  // per flavour and device: dynamic LDS the kernel has been allowed

// stack, cells: entries of `ebits` bits (frame_words x depth and 3 x cells of them per problem); input: long longs
extern "C" hipError_t pipk_launch_quast(const QProb *probs, const long long *input, void *stack, void *cells, int *out,
                                        int nprob, const QCaps *cap, int ebits, hipStream_t stream) {
  if (nprob <= 0) return hipSuccess;
  const size_t shm = pipk_quast_lds_bytes(cap, ebits);
  const int fl = ebits == 128 ? 1 : 0;
  const void *fn = fl ? (const void *)pip_quast_kernel<w128> : (const void *)pip_quast_kernel<w64>;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || (size_t)g_quast_lds[fl][dev] < shm) {  // opt in to more dynamic LDS, once per device and size
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) g_quast_lds[fl][dev] = (int)shm;
  }
  if (fl)
    hipLaunchKernelGGL(pip_quast_kernel<w128>, dim3(nprob), dim3(64), shm, stream, probs, input, (w128 *)stack, (w128 *)cells, out,
                       nprob, *cap);
  else
    hipLaunchKernelGGL(pip_quast_kernel<w64>, dim3(nprob), dim3(64), shm, stream, probs, input, (w64 *)stack, (w64 *)cells, out, nprob,
                       *cap);
  return hipGetLastError();
}
extern "C" hipError_t pipk_launch_quast_pack(const long long *cells, const long long *off, long long *packed, int nprob,
                                             int cells_cap, int ebits, hipStream_t stream) {
  if (nprob <= 0) return hipSuccess;
  hipLaunchKernelGGL(pip_quast_pack_kernel, dim3(nprob), dim3(128), 0, stream, cells, off, packed, cells_cap, ebits == 128 ? 6 : 3);
  return hipGetLastError();
}
