// piplib_amd/csrc/pip_advance.h -- the pivot kernel (pip_advance_kernel) and its device helpers.
//
// Included by pip_kernels.hip (which keeps the other kernels and every launcher) and by the
// pip_adv_*.hip translation units, each of which instantiates one group of the kernel's template
// instantiations (pip_adv_inst.h): the 28 instantiations compile side by side instead of one after
// the other.
#ifndef PIP_ADVANCE_H
#define PIP_ADVANCE_H
//
// One workgroup (256 threads = 4 wave64) owns one PIP problem ("job") and runs
// PipLib's dual-simplex pivot loop on it:
//
//   traiter()      reference source/traiter.c:628-791   -> pip_advance_kernel main loop
//   chercher()     traiter.c:39-44                      -> first-Minus search fused into phase C
//   exam_coef()    traiter.c:101-159                    -> exam_rows()   (from per-row sign summaries)
//   choisir_piv()  traiter.c:297-341                    -> choose_column() (wave 0, row-ordered tournament)
//   pivoter()      traiter.c:345-548                    -> phases A (wave 0), B (all waves), C (all threads)
//   integrer()     integrer.c:305-534 (constant cuts)   -> phase G
//   tab_sort_rows  traiter.c:556-623                    -> sort_rows()
//
// Data layout (all int64 "Entier" numerators, wrap-around arithmetic exactly as
// the reference's `long long` build):
//   * the tableau lives in HBM: S row slots of W int64 each (W even, 16-byte
//     aligned rows so a wave reads/writes a row with 16 B per lane, coalesced);
//   * logical row i is either a unit row (identity on column ref[i]) or a real
//     row stored in slot ref[i]; flags/denominators/ref of all logical rows are
//     staged in LDS for the whole solve, together with
//       - the pivot row,
//       - a per-row sign summary (so sign tests never touch HBM),
//       - a per-row non-zero bitmap (so the pivot-column tournament and the
//         elimination step only load rows that can matter);
//   * wave-level ballots / shuffles implement the pivot-column tournament, the
//     row-gcd refinement and all sign tests.  No MFMA: exact integer work.
//
// Rows whose pivot-column entry is zero and whose gcd with their denominator is
// already 1 are not rewritten: the reference multiplies them by 1, subtracts 0
// and divides by gcd 1 (traiter.c:470-501), i.e. leaves the same bits.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "pip_job.h"

typedef long long i64;
typedef unsigned long long u64;
typedef __int128 i128;
typedef unsigned __int128 u128;

// Entry ("Entier") type traits.  int64: the reference's `long long` build, two columns per
// lane and chunk (16 B); int128: the overflow-safe variant, one column per lane and chunk.
template <class T>
struct ET;
template <>
struct ET<i64> {
  typedef u64 U;
  static constexpr int BITS = 64, CPL = 2, EW = 1;
};
template <>
struct ET<i128> {
  typedef u128 U;
  static constexpr int BITS = 128, CPL = 1, EW = 2;
};
typedef unsigned short u16;
typedef unsigned char u8;

#ifndef PIP_OPT_CSMALL
#ifndef PIP_OPT_NARROW128
#define PIP_OPT_NARROW128 1  // (A/B switch) 128-bit rows of class 0: the row update on 64-bit registers
#endif
#define PIP_OPT_CSMALL 1   // (A/B switch) choisir_piv with 24-bit cross products while every row is in class 0
#endif
#ifndef PIP_OPT_INV_REDUCE
#define PIP_OPT_INV_REDUCE 1  // (A/B switch) row gcd and division by inverse multiplication (reduce_by_inverse) instead of remainders
#endif
#ifndef PIP_OPT_LANEPREP
#define PIP_OPT_LANEPREP 1  // (A/B switch) 128-bit rows: gcds and multipliers of a pivot's rows prepared one row per lane
#endif
#ifndef PIP_MINWAVES128
#define PIP_MINWAVES128 4  // waves per SIMD the 128-bit kernels of <= 256 columns are bounded to (128 VGPRs; 1 = no bound)
#endif
#ifndef PIP_MINWAVES
#define PIP_MINWAVES 6
#endif
// Diagnostic builds only (tools/pmc_dup.sh): -DPIP_DUP=n executes one idempotent piece of the
// pivot loop twice, so that the difference of the SQ_INSTS_* counters against the normal build
// is that piece's dynamic instruction count.  0 = off (every shipped/timed build).
#ifndef PIP_DUP
#define PIP_DUP 0
#endif
#define PIP_DUP_REPS(n) ((PIP_DUP == (n)) ? 2 : 1)
#define PIP_OPAQUE_MEM() asm volatile("" ::: "memory")
#define BIG_I 0x7fffffff
#define NOROW 0xffff

#ifdef PIP_PROFILE_EVENTS  // second diagnostic build: event counters (their atomics distort the cycle stamps)
__device__ unsigned long long *pf_buf;
#define CNT(i, n)                                                                          \
  do {                                                                                     \
    if ((threadIdx.x & 63) == 0 && pf_buf) atomicAdd(&pf_buf[16 + (i)], (unsigned long long)(n)); \
  } while (0)
#else
#define CNT(i, n)
#endif

// ---------------------------------------------------------------- integer ops
// piplib.h:128-169 + integrer.c:43-74 on wrap-around 64-bit integers.
// the wave's ballot as the compare instruction itself (hip's __ballot first materialises the predicate as 0 / 1 in a
// vector register and compares that: two more VALU instructions per ballot)
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
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
  CNT(7, 1);
  do {
    CNT(8, 1);
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
__device__ __forceinline__ unsigned gcd_u32(unsigned a, unsigned b) {
  if (a == 0) return b;
  if (b == 0) return a;
  int sh = __builtin_ctz(a | b);
  a >>= __builtin_ctz(a);
  CNT(9, 1);
  do {
    CNT(10, 1);
    b >>= __builtin_ctz(b);
    if (a > b) {
      unsigned t = a;
      a = b;
      b = t;
    }
    b -= a;
  } while (b);
  return a << sh;
}
__device__ __forceinline__ u64 gcd_mag(u64 a, u64 b) {
  if (a == 1 || b == 1) return 1;
  if (((a | b) >> 32) == 0) return gcd_u32((unsigned)a, (unsigned)b);
  return gcd_u64(a, b);
}
__device__ __forceinline__ i64 gcd_i64(i64 a, i64 b) { return (i64)gcd_mag(uabs64(a), uabs64(b)); }
// wave-uniform values: pin them to scalar registers so that the gcd / division / inverse
// chains that follow run on the scalar unit instead of occupying all 64 vector lanes
__device__ __forceinline__ i64 uni64(i64 v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u64)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((u64)v >> 32));
  return (i64)(((u64)hi << 32) | lo);
}
__device__ __forceinline__ i64 readlane64(i64 v, int src) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)(u64)v, src);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)((u64)v >> 32), src);
  return (i64)(((u64)hi << 32) | lo);
}

// C '/' and '%' made total (the CPU traps on x / 0 and MIN / -1).
__device__ __forceinline__ i64 cquo(i64 a, i64 b) {
  if (b == 1) return a;
  if (b == 0) return 0;
  if (b == -1) return wneg(a);
  if ((i64)(int)a == a && (i64)(int)b == b) {
    CNT(11, 1);
    return (i64)((int)a / (int)b);
  }
  CNT(12, 1);
  return a / b;
}
__device__ __forceinline__ i64 crem(i64 a, i64 b) {
  if (b == 0 || b == -1 || b == 1) return 0;
  if ((i64)(int)a == a && (i64)(int)b == b) return (i64)((int)a % (int)b);
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
__device__ __forceinline__ int bitlen64(u64 u) { return u ? 64 - __builtin_clzll(u) : 0; }
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

// ---------------------------------------------------------------- 128-bit counterparts
__device__ __forceinline__ u128 uabs64(i128 x) { return x < 0 ? (u128)0 - (u128)x : (u128)x; }
__device__ __forceinline__ i128 wmul(i128 a, i128 b) { return (i128)((u128)a * (u128)b); }
__device__ __forceinline__ i128 wsub(i128 a, i128 b) { return (i128)((u128)a - (u128)b); }
__device__ __forceinline__ i128 wadd(i128 a, i128 b) { return (i128)((u128)a + (u128)b); }
__device__ __forceinline__ i128 wneg(i128 a) { return (i128)((u128)0 - (u128)a); }
__device__ __forceinline__ int ctz128(u128 x) {
  u64 lo = (u64)x;
  return lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll((u64)(x >> 64));
}
__device__ __forceinline__ int bitlen64(u128 x) {
  u64 hi = (u64)(x >> 64);
  return hi ? 128 - __builtin_clzll(hi) : bitlen64((u64)x);
}
__device__ __forceinline__ bool fits64(i128 x) { return (i128)(i64)x == x; }
// a mod b for a 32-bit b != 0, one 32-bit digit of a at a time: t = r*2^32 + digit < b*2^32,
// so the quotient of each step is below 2^32 and a double-precision estimate of it is off by
// at most one.
__device__ __forceinline__ unsigned umod128_32(u128 a, unsigned b) {
  unsigned r = 0;
#pragma unroll
  for (int k = 3; k >= 0; k--) {
    const u64 t = ((u64)r << 32) | (unsigned)(a >> (32 * k));
    const u64 q = (u64)((double)t / (double)b);
    i64 d = (i64)(t - q * b);
    if (d < 0) d += b;
    if (d >= (i64)b) d -= b;
    r = (unsigned)d;
  }
  return r;
}
__device__ __forceinline__ u128 gcd_mag(u128 a, u128 b) {
  if (((a | b) >> 64) == 0) return gcd_mag((u64)a, (u64)b);
  if (a == 0) return b;
  if (b == 0) return a;
  // a wide determinant limb against a small denominator: one Euclid step first -- the
  // subtractive binary gcd below needs a 128-bit iteration per bit of the size difference
  if ((b >> 32) == 0) return (u128)gcd_mag((u64)(unsigned)b, (u64)umod128_32(a, (unsigned)b));
  if ((a >> 32) == 0) return (u128)gcd_mag((u64)(unsigned)a, (u64)umod128_32(b, (unsigned)a));
  int sh = ctz128(a | b);
  a >>= ctz128(a);
  CNT(20, 1);
  do {
    CNT(21, 1);
    b >>= ctz128(b);
    if (a > b) {
      u128 t = a;
      a = b;
      b = t;
    }
    b -= a;
  } while (b);
  return a << sh;
}
__device__ __forceinline__ i128 gcd_i64(i128 a, i128 b) { return (i128)gcd_mag(uabs64(a), uabs64(b)); }
__device__ __forceinline__ u128 inv_odd64(u128 m) {
  // the inverse modulo 2^64 on 64-bit registers (five Newton steps from 3 correct bits), one 128-bit step on top:
  // a 128-bit multiplication is ten 32-bit multiply-adds, a 64-bit one three
  u128 x = (u128)inv_odd64((u64)m);
  x *= 2 - m * x;
  return x;
}
// every division on the pivot path is exact (piplib_int_div_exact of a gcd): shift + odd inverse
__device__ __forceinline__ i128 cquo(i128 a, i128 b) {
  if (b == 1) return a;
  if (b == 0) return 0;
  if (fits64(a) && fits64(b)) return (i128)cquo((i64)a, (i64)b);
  const bool neg = b < 0;
  u128 ub = uabs64(b);
  int s = ctz128(ub);
  i128 q = (i128)((u128)(a >> s) * inv_odd64(ub >> s));
  return neg ? wneg(q) : q;
}
__device__ __forceinline__ u128 umod128(u128 a, u128 g) {
  if (((a | g) >> 64) == 0) return (u64)a % (u64)g;
  if (a < g) return a;
  u128 rem = 0;
  for (int i = bitlen64(a) - 1; i >= 0; i--) {
    rem = (rem << 1) | ((a >> i) & 1);
    if (rem >= g) rem -= g;
  }
  return rem;
}
__device__ __forceinline__ i128 crem(i128 a, i128 b) {
  if (b == 0 || b == -1 || b == 1) return 0;
  if (fits64(a) && fits64(b)) return (i128)crem((i64)a, (i64)b);
  u128 r = umod128(uabs64(a), uabs64(b));
  return a < 0 ? wneg((i128)r) : (i128)r;  // C remainder: sign of the dividend
}
__device__ __forceinline__ i128 fmod64(i128 a, i128 b) {
  i128 m = crem(a, b);
  if (m < 0) m = wadd(m, (i128)uabs64(b));
  return m;
}
__device__ __forceinline__ int log2_64(i128 x) {
  u128 u = uabs64(x);
  return u == 0 ? 1 : bitlen64(u);
}
__device__ __forceinline__ int sign_code(i128 x) { return x > 0 ? 1 : (x < 0 ? 2 : 0); }
__device__ __forceinline__ i128 readlane64(i128 v, int src) {
  u64 lo = (u64)readlane64((i64)(u64)(u128)v, src), hi = (u64)readlane64((i64)(u64)((u128)v >> 64), src);
  return (i128)(((u128)hi << 64) | lo);
}
__device__ __forceinline__ i128 uni64(i128 v) {
  u64 lo = (u64)uni64((i64)(u64)(u128)v), hi = (u64)uni64((i64)(u64)((u128)v >> 64));
  return (i128)(((u128)hi << 64) | lo);
}
// a / d for d = a positive gcd that divides a: when both fit 32 bits, shift + 32-bit odd inverse
// (four single multiplies) instead of a division; two quotients by the same d share the inverse.
template <class T>
__device__ __forceinline__ T exact_quo(T a, T d) {
  const auto ua = uabs64(a);
  if (((ua | (decltype(ua))d) >> 32) == 0) {
    unsigned m = (unsigned)d;
    const int sh = __builtin_ctz(m);
    m >>= sh;
    unsigned inv = m;
    inv *= 2u - m * inv;
    inv *= 2u - m * inv;
    inv *= 2u - m * inv;
    inv *= 2u - m * inv;
    const unsigned q = ((unsigned)ua >> sh) * inv;
    return a < 0 ? wneg((T)q) : (T)q;
  }
  return cquo(a, d);
}

__device__ __forceinline__ u128 umod_small(u128 a, u128 g, bool small32) {
  if (small32) return (u128)((unsigned)a % (unsigned)g);
  return umod128(a, g);
}
__device__ __forceinline__ u64 umod_small(u64 a, u64 g, bool small32) {
  return small32 ? (u64)((unsigned)a % (unsigned)g) : a % g;
}
__device__ __forceinline__ double to_double(i64 x) { return (double)x; }
__device__ __forceinline__ double to_double(i128 x) {
  const u128 m = uabs64(x);  // via the magnitude: hi*2^64 + lo on a negative value would cancel
  const double d = (double)(u64)(m >> 64) * 18446744073709551616.0 + (double)(u64)m;
  return x < 0 ? -d : d;
}
// workgroup barrier; a single-wave workgroup only needs the compiler to keep LDS order
template <int NW>
__device__ __forceinline__ void bsync() {
  if (NW > 1)
    __syncthreads();
  else
    __builtin_amdgcn_wave_barrier();
}

// per-row sign summary kept in LDS (exam_coef and the post-pivot flag update read only this)
//  bits 0-1 constant term, bit 2 some parameter coef > 0, bit 3 some < 0,
//  bits 4-5 big-parameter coef, bits 6-7 coef in the column just pivoted on.
#define SIG_CONST(s) ((s)&3)
#define SIG_PPOS(s) (((s) >> 2) & 1)
#define SIG_PNEG(s) (((s) >> 3) & 1)
#define SIG_BIG(s) (((s) >> 4) & 3)
#define SIG_PIV(s) (((s) >> 6) & 3)
#define SIG_RED 256  // gcd(row, denominator) is known to be 1 (the row needs no reduction)
#define UNITBIT 0x8000
#define UNITZERO 0x4000  // a unit row flagged Unit|Zero (it was a pivot row, traiter.c:514); else just Unit
#define UNITCOL(rf) ((rf)&0x3ff)

// Optional phase profile (diagnostic build only: -DPIP_PROFILE; never shipped/timed).
#ifdef PIP_PROFILE
#define PROF_DECL u64 pf_t = __builtin_readcyclecounter(), pf_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PROF(i)                                \
  do {                                         \
    u64 pf_n = __builtin_readcyclecounter();   \
    pf_acc[i] += pf_n - pf_t;                  \
    pf_t = pf_n;                               \
  } while (0)
#ifdef PIP_PROFILE_EVENTS
#define PROF_CNT(buf, i, c)                                                     \
  do {                                                                          \
    if ((threadIdx.x & 63) == 0 && buf && (c)) atomicAdd(&buf[16 + (i)], 1ull); \
  } while (0)
#else
#define PROF_CNT(buf, i, c)
#endif
#define PROF_FLUSH(buf)                                                \
  do {                                                                 \
    if (threadIdx.x == 0 && buf)                                       \
      for (int q_ = 0; q_ < 16; q_++) atomicAdd(&buf[q_], pf_acc[q_]); \
  } while (0)
#else
#define PROF_DECL
#define PROF(i)
#define PROF_CNT(buf, i, c)
#define PROF_FLUSH(buf)
#endif

template <class T>
__device__ __forceinline__ int colof(int c, int lane, int h) {
  return c * (64 * ET<T>::CPL) + ET<T>::CPL * lane + h;
}
__device__ __forceinline__ int ctzU(u64 x) { return __builtin_ctzll(x); }
__device__ __forceinline__ int ctzU(u128 x) { return ctz128(x); }
// magnitude classes: every entry of a class-c row is below 2^cls_bits(c)
template <class T>
__device__ __forceinline__ int cls_bits(int c) {
  return c < 3 ? (ET<T>::BITS / 4) * (c + 1) - 1 : ET<T>::BITS;
}
template <class T>
__device__ __forceinline__ int cls_of(typename ET<T>::U orall) {  // wave-collective: class of the OR of all lanes
  constexpr int B4 = ET<T>::BITS / 4;
  return ballot64((orall >> (3 * B4 - 1)) != 0) ? 3
         : (ballot64((orall >> (2 * B4 - 1)) != 0) ? 2 : (ballot64((orall >> (B4 - 1)) != 0) ? 1 : 0));
}

// LDS image of one job.  L = logical rows, S = row slots (real rows), WP = NCH*128 columns,
// NM = 2*NCH mask words per row.  Column j of a row is owned by lane (j%128)/2 of the wave
// that holds the row, register (c = j/128, h = j&1); a row's non-zero bitmap uses the same
// geometry: word 2c+h, bit (j%128)/2.  Everything that only real rows have is indexed by
// slot, so the per-pivot loops run over the real rows only.
// bytes of the LDS region shared by prow and the entry-time sort keys (Smax floats)
__host__ __device__ __forceinline__ size_t prow_bytes(size_t prow, int Smax) {
  size_t k = sizeof(float) * (size_t)Smax;
  return ((prow > k ? prow : k) + 15) & ~(size_t)15;
}

template <class T>
struct Shared {
  T *den;     // [S]  denominator of the row in slot s
  T *prow;      // [WP] pivot row (zero beyond ncol)
  T *cst;       // [S]  constant term (column nvar) of the row in slot s
  u64 *nzm;     // [S][NM] non-zero bitmap
  float *size;  // [S]  tab_sort_rows key (entry only)
  u16 *sig;     // [S]  sign summary
  u16 *srow;    // [S]  slot -> logical row
  u16 *work;    // [S]  slots the current pivot rewrites
  u16 *ref;     // [L]  logical row -> slot, or UNITBIT | (UNITZERO) | column for a unit row
  u16 *urow;    // [WP] unknown column -> logical row of its unit row
  u8 *fl;       // [S]  flag of the row in slot s (Plus/Minus/Zero/Critic/Unknown)
  u8 *nf;       // [S]  flag exam_coef would give an Unknown row
  u8 *rcls;     // [S]  magnitude class of the row's largest entry (see CLS_BITS)
};

struct Scalars {
  int pivi, pivi2, pivj, tmp, tmp2, aux;
  int flagor, nwork, bad, ovf;
  u64 smaxbits;
};

template <class T, int NCH>
struct RowRegs {
  T v[NCH][ET<T>::CPL];
};

// ---- coalesced row access, 16 B per lane: lane l of a wave owns columns colof(c, l, h) ----
template <class T, int NCH>
__device__ __forceinline__ void row_load(RowRegs<T, NCH> &r, const T *row, int ncolp, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    int j0 = colof<T>(c, lane, 0);
    if (j0 < ncolp) {
      const longlong2 t = *reinterpret_cast<const longlong2 *>(row + j0);
      if constexpr (ET<T>::CPL == 2) {
        r.v[c][0] = t.x;
        r.v[c][1] = t.y;
      } else {
        r.v[c][0] = (T)(((u128)(u64)t.y << 64) | (u64)t.x);
      }
    } else {
#pragma unroll
      for (int h = 0; h < ET<T>::CPL; h++) r.v[c][h] = 0;
    }
  }
}
template <class T, int NCH>
__device__ __forceinline__ void row_store(const RowRegs<T, NCH> &r, T *row, int ncolp, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    int j0 = colof<T>(c, lane, 0);
    if (j0 < ncolp) {
      longlong2 t;
      if constexpr (ET<T>::CPL == 2) {
        t.x = r.v[c][0];
        t.y = r.v[c][1];
      } else {
        t.x = (i64)(u64)(u128)r.v[c][0];
        t.y = (i64)(u64)((u128)r.v[c][0] >> 64);
      }
      *reinterpret_cast<longlong2 *>(row + j0) = t;
    }
  }
}

// select the register that holds column pivj (uniform c,h) and read it from its owner lane
template <class T, int NCH>
__device__ __forceinline__ T row_entry(const RowRegs<T, NCH> &r, int pc, int ph, int pl) {
  T mine = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < ET<T>::CPL; h++)
      if (c == pc && h == ph) mine = r.v[c][h];
  return readlane64(mine, pl);
}

// Sign summary, non-zero bitmap and magnitude class of a row held in registers
// (wave-collective).  Lane 0 publishes them for slot s.
template <class T, int NCH>
__device__ __forceinline__ void row_publish(const RowRegs<T, NCH> &r, const Shared<T> &S, int s, int nvar, int ncol,
                                            int bigparm, int pivj, int extra_sig, bool has_parm, int lane) {
  int bs = 0;
  bool ppos = false, pneg = false;
  typename ET<T>::U mx = 0;
  u64 nz[NCH * ET<T>::CPL];
  // the constant term and the entry in the column just pivoted on: read from their owner lanes
  // into scalars (cheaper than per-lane sign codes and ballots)
  constexpr int CW = 64 * ET<T>::CPL;
  const T cz = row_entry<T, NCH>(r, nvar / CW, nvar % ET<T>::CPL, (nvar % CW) / ET<T>::CPL);
  const int cs = sign_code(cz);
  int ps = 0;
  if (pivj >= 0) ps = sign_code(row_entry<T, NCH>(r, pivj / CW, pivj % ET<T>::CPL, (pivj % CW) / ET<T>::CPL));
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < ET<T>::CPL; h++) {
      int j = colof<T>(c, lane, h);
      T z = r.v[c][h];
      mx |= uabs64(z);
      if (has_parm) {
        if (j == bigparm) bs = sign_code(z);
        if (j > nvar && j < ncol) {
          ppos |= z > 0;
          pneg |= z < 0;
        }
      }
      nz[ET<T>::CPL * c + h] = ballot64(z != 0);
    }
  int sig = extra_sig | cs | (ps << 6);
  if (has_parm) {
    sig |= (ballot64(ppos) ? 4 : 0) | (ballot64(pneg) ? 8 : 0);
    sig |= (ballot64(bs == 1) ? 16 : 0) | (ballot64(bs == 2) ? 32 : 0);
  }
  const int cls = cls_of<T>(mx);
  if (lane == 0) {
    S.sig[s] = (u16)sig;
    S.rcls[s] = (u8)cls;
    S.cst[s] = cz;
#pragma unroll
    for (int e = 0; e < NCH * ET<T>::CPL; e++) S.nzm[(size_t)s * (NCH * ET<T>::CPL) + e] = nz[e];
  }
}

// ---- rows whose entries all fit an int (64-bit tableaux): the same three steps on 32-bit registers ----
template <int NCH>
struct RowRegs32 {
  int v[NCH][2];
};
template <int NCH>
__device__ __forceinline__ void row_store32(const RowRegs32<NCH> &z, i64 *row, int ncolp, int lane) {
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int j0 = colof<i64>(c, lane, 0);
    if (j0 < ncolp) {
      longlong2 t;
      t.x = (i64)z.v[c][0];
      t.y = (i64)z.v[c][1];
      *reinterpret_cast<longlong2 *>(row + j0) = t;
    }
  }
}
template <int NCH>
__device__ __forceinline__ int row_entry32(const RowRegs32<NCH> &z, int pc, int ph, int pl) {
  int mine = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++)
      if (c == pc && h == ph) mine = z.v[c][h];
  return __builtin_amdgcn_readlane(mine, pl);
}
// row_publish for a row without parameter columns (nparm == 0) held as ints; returns its class (0: every entry
// below 2^15, 1: below 2^31)
template <int NCH>
__device__ __forceinline__ int row_publish32(const RowRegs32<NCH> &z, const Shared<i64> &S, int s, int nvar, int pivj,
                                             int extra_sig, int lane) {
  const int cz = row_entry32<NCH>(z, nvar / 128, nvar % 2, (nvar % 128) / 2);
  int sig = extra_sig | (cz > 0 ? 1 : (cz < 0 ? 2 : 0));
  if (pivj >= 0) {
    const int pz = row_entry32<NCH>(z, pivj / 128, pivj % 2, (pivj % 128) / 2);
    sig |= (pz > 0 ? 1 : (pz < 0 ? 2 : 0)) << 6;
  }
  unsigned mx = 0;
  u64 nz[NCH * 2];
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int v = z.v[c][h];
      mx |= (unsigned)(v < 0 ? -v : v);
      nz[2 * c + h] = ballot64(v != 0);
    }
  const int cls = ballot64((mx >> 15) != 0) ? 1 : 0;
  if (lane == 0) {
    S.sig[s] = (u16)sig;
    S.rcls[s] = (u8)cls;
    S.cst[s] = (i64)cz;
#pragma unroll
    for (int e = 0; e < NCH * 2; e++) S.nzm[(size_t)s * (NCH * 2) + e] = nz[e];
  }
  return cls;
}

// Wave-wide unsigned min / max without LDS traffic: a DPP butterfly inside each row of 16
// lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row results
// through the scalar unit.  Call with all 64 lanes active; the result is wave-uniform.
template <bool MAX>
__device__ __forceinline__ unsigned wave_minmax_u32(unsigned v) {
#define PIP_DPP_STEP(ctrl)                                                                         \
  {                                                                                                \
    unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, false);     \
    v = MAX ? (o > v ? o : v) : (o < v ? o : v);                                                   \
  }
  PIP_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
  PIP_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]
  PIP_DPP_STEP(0x141)  // row_half_mirror
  PIP_DPP_STEP(0x140)  // row_mirror
#undef PIP_DPP_STEP
  const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  if (MAX) {
    const unsigned a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
  }
  const unsigned a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
  return a < b ? a : b;
}

// Row gcd and exact division of a row without remainders.  The refinement of row_reduce_rem (below) reduces every entry
// modulo the current g, round after round -- a division sequence per entry and round, for 128-bit operands a bit-serial
// loop of up to 128 steps.  Here one wrap-around multiplication per entry answers "does g divide it" AND yields the
// quotient: with g = 2^s m (m odd) and inv = m^-1 mod 2^W, an entry a whose low s bits are zero has a >> s = k m  <=>
// (a >> s) inv mod 2^W = k, and the quotients of non-multiples are the values above (2^W - 1) / m (multiplication by inv
// is a bijection that maps the multiples of m onto 0 .. (2^W - 1) / m).  Every |a| is below 2^B (B = bits of the OR of the
// row's magnitudes), so a true quotient is below 2^(B - s - bits(m) + 1) and a false one at least 2^(W - bits(m)): one
// shift tells them apart.  (With signs: entries lie in (-2^B, 2^B) and the quotient is read as a signed number, so a
// false quotient k needs |k m| >= 2^W - 2^B while an accepted one has |k m| < 2^(B - s + 1): B <= W - 2 keeps them apart.)  A round that meets a non-multiple folds that entry into g and starts
// again.  The gcd is associative, so the result is the reference's left fold (traiter.c:479-501).
// W is the narrowest of 32 / 64 / 128 bits that holds the entries and g (WT<TW>), whatever the row's entry type.
template <class TW>
struct WT;
template <>
struct WT<int> {
  typedef unsigned U;
};
template <>
struct WT<i64> {
  typedef u64 U;
};
template <>
struct WT<i128> {
  typedef u128 U;
};
__device__ __forceinline__ int ctzW(unsigned x) { return __builtin_ctz(x); }
__device__ __forceinline__ int ctzW(u64 x) { return __builtin_ctzll(x); }
__device__ __forceinline__ int ctzW(u128 x) { return ctz128(x); }
__device__ __forceinline__ int bitlenW(unsigned x) { return x ? 32 - __builtin_clz(x) : 0; }
__device__ __forceinline__ int bitlenW(u64 x) { return bitlen64(x); }
__device__ __forceinline__ int bitlenW(u128 x) { return bitlen64(x); }
__device__ __forceinline__ unsigned invW(unsigned m) {
  unsigned x = (3u * m) ^ 2u;  // 5 correct bits, doubled by every step
  x *= 2u - m * x;
  x *= 2u - m * x;
  x *= 2u - m * x;
  return x;
}
__device__ __forceinline__ u64 invW(u64 m) { return inv_odd64(m); }
__device__ __forceinline__ u128 invW(u128 m) { return inv_odd64(m); }
__device__ __forceinline__ unsigned gcdW(unsigned a, unsigned b) { return gcd_u32(a, b); }
__device__ __forceinline__ u64 gcdW(u64 a, u64 b) { return gcd_mag(a, b); }
__device__ __forceinline__ u128 gcdW(u128 a, u128 b) { return gcd_mag(a, b); }
__device__ __forceinline__ unsigned uabsW(int x) { return x < 0 ? 0u - (unsigned)x : (unsigned)x; }
__device__ __forceinline__ u64 uabsW(i64 x) { return uabs64(x); }
__device__ __forceinline__ u128 uabsW(i128 x) { return uabs64(x); }
__device__ __forceinline__ int readlaneW(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ i64 readlaneW(i64 v, int src) { return readlane64(v, src); }
__device__ __forceinline__ i128 readlaneW(i128 v, int src) { return readlane64(v, src); }

// zz (N entries per lane, every |zz| below 2^B, B at most W - 2) is divided by gcd(g, all |zz|) in place; returns that
// gcd, with its factor 2^s and the inverse of its odd part (the caller divides the denominator with them).  g >= 1.
// Returns 0 -- zz as it came -- if a round made no progress (it cannot; a wave must never spin on it): the caller falls
// back to the remainder loop.
// INPLACE: the quotients take the entries' registers and a failing round undoes its multiplications (times m: the same
// bijection backwards) -- for 128-bit entries and for callers short of registers.
template <class TW, int N, bool INPLACE = (sizeof(TW) > 8)>
__device__ __forceinline__ typename WT<TW>::U reduce_by_inverse(TW (&zz)[N], int B, typename WT<TW>::U g, int &s, typename WT<TW>::U &inv) {
  typedef typename WT<TW>::U U;
  for (;;) {
    s = 0;
    inv = 1;
    if (g == 1) return g;
    s = ctzW(g);
    const U m = g >> s;
    if (s) {  // the low s bits of every entry (two's complement: those of z are zero iff those of |z| are)
      const U lowmask = ((U)1 << s) - 1;
      TW pick = 0;
#pragma unroll
      for (int e = 0; e < N; e++)
        if (((U)zz[e] & lowmask) != 0) pick = zz[e];
      const u64 bad = ballot64(pick != 0);
      if (bad) {
        const U g2 = gcdW(g, uabsW(readlaneW(pick, __ffsll((long long)bad) - 1)));
        if (g2 == g) return 0;  // (cannot be: an entry with low bits set is no multiple of 2^s) -- never loop on it
        g = g2;
        continue;
      }
    }
    if (m == 1) {  // a power of two
#pragma unroll
      for (int e = 0; e < N; e++) zz[e] >>= s;
      return g;
    }
    int lim = B - s - bitlenW(m) + 1;
    lim = lim < 0 ? 0 : lim;
    CNT(13, 1);
    inv = invW(m);
    if constexpr (!INPLACE) {
      TW qq[N], pick = 0;
#pragma unroll
      for (int e = 0; e < N; e++) {
        const TW q = (TW)((U)(zz[e] >> s) * inv);
        qq[e] = q;
        if ((uabsW(q) >> lim) != 0) pick = zz[e];  // (a failing entry is not zero: zero's quotient is zero)
      }
      const u64 bad = ballot64(pick != 0);
      if (!bad) {
#pragma unroll
        for (int e = 0; e < N; e++) zz[e] = qq[e];
        return g;
      }
      CNT(14, 1);
      const U g2 = gcdW(g, uabsW(readlaneW(pick, __ffsll((long long)bad) - 1)));
      if (g2 == g) return 0;  // (cannot be, see the proof above: a rejected entry is no multiple of g) -- never loop on it
      g = g2;
    } else {
      bool okl = true;
#pragma unroll
      for (int e = 0; e < N; e++) {
        const TW q = (TW)((U)(zz[e] >> s) * inv);
        zz[e] = q;
        okl &= (uabsW(q) >> lim) == 0;
      }
      const u64 bad = ballot64(!okl);
      if (!bad) return g;
      CNT(14, 1);
      TW pick = 0;
#pragma unroll
      for (int e = 0; e < N; e++) {
        const TW q = zz[e];
        const TW a = (TW)(((U)q * m) << s);
        zz[e] = a;
        if ((uabsW(q) >> lim) != 0) pick = a;
      }
      const U g2 = gcdW(g, uabsW(readlaneW(pick, __ffsll((long long)bad) - 1)));
      if (g2 == g) return 0;
      g = g2;
    }
  }
}

// the denominator's share: newden = g0 / g for the g = 2^s m reduce_by_inverse returned (inv = m^-1 modulo its width)
template <class T, class TW>
__device__ __forceinline__ T reduce_den(T g0, typename WT<TW>::U g, int s, typename WT<TW>::U inv) {
  typedef typename ET<T>::U U;
  if (g == 1) return g0;
  if constexpr (sizeof(TW) == sizeof(T)) {
    return (T)((U)(g0 >> s) * (U)inv);
  } else {
    if ((T)(TW)g0 == g0) return (T)(TW)((typename WT<TW>::U)((TW)g0 >> s) * inv);  // the quotient of a narrow g0 is narrow
    return (T)((U)(g0 >> s) * inv_odd64((U)(g >> s)));
  }
}

template <class T, int N>
__device__ __forceinline__ bool row_reduce_rem(T (&z)[N], typename ET<T>::U mx, T g0, int lane, T &newden);

// pivoter()'s row gcd and division (traiter.c:479-501): z (N entries per lane, `mx` = OR of the lane's |z|) is divided by
// g = gcd(g0, z_0, ..., z_n) in place; newden = g0 / g.  False where the reference would divide by zero.
// `zfold`: an entry of the row the caller knows beforehand (the one in the pivot column; 0: none) -- it is folded into g
// first, after which g is the row's gcd more often than not and one round does it; `gpre` != 0: gcd(|g0|, |zfold|) as
// the caller has it already (the lane-parallel preparation of the 128-bit phase B).
// TRY32 = false: no 32-bit variant (callers that are short of registers).
template <class T, int N, bool TRY32 = true>
__device__ __forceinline__ bool row_reduce(T (&z)[N], typename ET<T>::U mx, T g0, int lane, T &newden, T zfold = 0,
                                           typename ET<T>::U gpre = 0) {
  typedef typename ET<T>::U U;
  newden = g0;
  if (g0 == 1) return true;
  const int B = (int)wave_minmax_u32<true>((unsigned)bitlen64(mx));
  // (g0 == 0 -- the reference would divide by zero unless some entry is not -- and entries of 2^(W-2) and more in magnitude
  // keep the remainder loop)
  if (!PIP_OPT_INV_REDUCE || g0 == 0 || B > ET<T>::BITS - 2) return row_reduce_rem<T, N>(z, mx, g0, lane, newden);
  U g = (U)uni64((T)uabs64(g0));
  if (gpre != 0)
    g = gpre;
  else if (zfold != 0)
    g = gcd_mag(g, (U)uni64((T)uabs64(zfold)));
  if (g == 1) return true;
  const int gb = bitlen64(g);
  if (TRY32 && B <= 30 && gb <= 32) {
    int zz[N], s;
    unsigned inv;
#pragma unroll
    for (int e = 0; e < N; e++) zz[e] = (int)z[e];
    const unsigned gg = reduce_by_inverse<int, N>(zz, B, (unsigned)g, s, inv);
    if (gg == 0) return row_reduce_rem<T, N>(z, mx, g0, lane, newden);
    if (gg != 1) {
#pragma unroll
      for (int e = 0; e < N; e++) z[e] = (T)zz[e];
    }
    newden = reduce_den<T, int>(g0, gg, s, inv);
    return true;
  }
  if constexpr (sizeof(T) == 16) {
    if (B <= 62 && gb <= 64) {
      i64 zz[N];
      int s;
      u64 inv;
#pragma unroll
      for (int e = 0; e < N; e++) zz[e] = (i64)z[e];
      const u64 gg = reduce_by_inverse<i64, N>(zz, B, (u64)g, s, inv);
      if (gg == 0) return row_reduce_rem<T, N>(z, mx, g0, lane, newden);
      if (gg != 1) {
#pragma unroll
        for (int e = 0; e < N; e++) z[e] = (T)zz[e];
      }
      newden = reduce_den<T, i64>(g0, gg, s, inv);
      return true;
    }
  }
  int s;
  U inv;
  const U gg = reduce_by_inverse<T, N, (sizeof(T) > 8 || !TRY32)>(z, B, g, s, inv);
  if (gg == 0) return row_reduce_rem<T, N>(z, mx, g0, lane, newden);
  newden = reduce_den<T, T>(g0, gg, s, inv);
  return true;
}

// pivoter()'s inner loop for one row (traiter.c:470-501), one wave per row.
//   z_j = p_j*lpiv - q_j*foo  (j != pivj),  z_pivj = dpiv*foo
//   g   = gcd(lpiv*den, z_0, ..., z_{ncol-1});  row /= g; den = lpiv*den/g
// The reference folds the gcd left to right and stops calling gcd once it hits
// 1; gcd is associative, so any evaluation order gives the same g.  We refine
// g downwards: reduce every z modulo the current g, fold in one non-zero
// remainder, repeat until all remainders vanish (typically <= 2 rounds).
// The second half: z (N entries per lane, `mx` = OR of the lane's |z|) is divided by g = gcd(g0, z_0, ..., z_n) in place;
// newden = g0 / g.  False where the reference would divide by zero.
template <class T, int N>
__device__ __forceinline__ bool row_reduce_rem(T (&z)[N], typename ET<T>::U mx, T g0, int lane, T &newden) {
  typedef typename ET<T>::U U;
  newden = g0;
  if (g0 == 1) return true;
  U g = (U)uni64((T)uabs64(g0));
  // 32-bit remainders when everything fits (the common case): one v_rcp-based
  // division instead of the 64-bit software routine
  const bool small = (ballot64((mx >> 32) != 0) == 0) && (g >> 32) == 0;
  for (;;) {
    U rr = 0;
#pragma unroll
    for (int e = 0; e < N; e++) {
      U a = uabs64(z[e]);
      U m = g == 0 ? a : umod_small(a, g, small);
      rr = rr ? rr : m;
    }
    CNT(13, 1);
    CNT(14, small ? 0 : 1);
    u64 nz = ballot64(rr != 0);
    if (!nz) break;
    int src = __ffsll((long long)nz) - 1;
    U r0 = (U)readlane64((T)rr, src);
    g = gcd_mag(g, r0);
    if (g == 1) break;
  }
  if (g == 1) return true;
  if (g == 0) return false;  // the reference would divide by zero here
  if (small) {
    // every |z| and g fit 32 bits: the exact quotients do too, so a 32-bit odd inverse (four
    // Newton steps of single multiplies) replaces the 64-bit one
    const unsigned g32 = (unsigned)g;
    const int s = __builtin_ctz(g32);
    const unsigned m = g32 >> s;
    unsigned inv = m;  // 3 correct bits, doubled by every step
    inv *= 2u - m * inv;
    inv *= 2u - m * inv;
    inv *= 2u - m * inv;
    inv *= 2u - m * inv;
#pragma unroll
    for (int e = 0; e < N; e++) {
      const T zz = z[e];
      const unsigned q = ((unsigned)uabs64(zz) >> s) * inv;
      z[e] = zz < 0 ? wneg((T)q) : (T)q;
    }
    const unsigned qd = ((unsigned)uabs64(g0) >> s) * inv;
    newden = g0 < 0 ? wneg((T)qd) : (T)qd;
    return true;
  }
  int s = ctzU(g);
  U inv = inv_odd64(g >> s);
#pragma unroll
  for (int e = 0; e < N; e++) z[e] = (T)((U)(z[e] >> s) * inv);
  newden = (T)((U)(g0 >> s) * inv);
  return true;
}

// `narrow64` (128-bit entries only): every entry of the row and of the pivot row and both multipliers fit a long long --
// the products are 64 x 64 -> 128 bits (four 32-bit multiply-adds each instead of ten) and cannot wrap; same bits.
// LEANREG: the caller is short of registers (the one-wave 64-bit kernels under their 80-register bound) -- the remainder loop.
template <class T, int NCH, bool LEANREG = false>
__device__ __forceinline__ bool update_row(RowRegs<T, NCH> &r, const T *prow, int pivj, T lpiv, T foo, T dpiv, T g0,
                                           int lane, T &newden, typename ET<T>::U gpre = 0, bool narrow64 = false) {
  typedef typename ET<T>::U U;
  U mx = 0;
  if constexpr (sizeof(T) == 16) {
    if (narrow64) {
      const i64 lp64 = (i64)lpiv, foo64 = (i64)foo;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int j = colof<T>(c, lane, 0);
        const i64 p = (i64)r.v[c][0], q = (i64)prow[j];
        T z = (T)p * (T)lp64 - (T)q * (T)foo64;
        if (j == pivj) z = wmul(dpiv, foo);
        r.v[c][0] = z;
        mx |= uabs64(z);
      }
      return row_reduce<T, NCH>(reinterpret_cast<T(&)[NCH]>(r.v), mx, g0, lane, newden, wmul(dpiv, foo), gpre);
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < ET<T>::CPL; h++) {
      int j = colof<T>(c, lane, h);
      T q = prow[j];
      T z = wsub(wmul(r.v[c][h], lpiv), wmul(q, foo));
      if (j == pivj) z = wmul(dpiv, foo);
      r.v[c][h] = z;
      mx |= uabs64(z);
    }
  if constexpr (LEANREG)
    return row_reduce_rem<T, NCH * ET<T>::CPL>(reinterpret_cast<T(&)[NCH * ET<T>::CPL]>(r.v), mx, g0, lane, newden);
  else
    return row_reduce<T, NCH * ET<T>::CPL>(reinterpret_cast<T(&)[NCH * ET<T>::CPL]>(r.v), mx, g0, lane, newden, wmul(dpiv, foo), gpre);
}

// The same for 128-bit rows whose operands are all below 2^31 (the row and the pivot row in magnitude class 0, the
// multipliers and the pivot row's denominator below 2^31) under a denominator product g0 that fits 63 bits: every
// product is below 2^62 and every z below 2^63, so the update runs on 64-bit registers -- and, when the z's and g0 fit 32
// bits as well, on row_reduce's 32-bit remainders and quotients -- with the same bits as the 128-bit code.
template <int NCH>
__device__ __forceinline__ bool update_row_narrow(RowRegs<i128, NCH> &r, const i128 *prow, int pivj, i64 lpiv, i64 foo, i64 dpiv,
                                                  i64 g0, int lane, i128 &newden, u64 gpre = 0) {
  i64 z[NCH];
  u64 mx = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int j = colof<i128>(c, lane, 0);
    const i64 p = (i64)r.v[c][0], q = (i64)prow[j];
    i64 v = p * lpiv - q * foo;
    if (j == pivj) v = dpiv * foo;
    z[c] = v;
    mx |= uabs64(v);
  }
  i64 nd;
  const bool ok = row_reduce<i64, NCH>(z, mx, g0, lane, nd, dpiv * foo, gpre);
#pragma unroll
  for (int c = 0; c < NCH; c++) r.v[c][0] = (i128)z[c];
  newden = (i128)nd;
  return ok;
}

// a mod g for a, g < 2^20, g >= 1, rg = v_rcp_f32(g): the float quotient estimate is off by at most one either way
// (a / g < 2^20, relative error of the reciprocal and of the product below 2^-22), so the remainder estimate is
// r - g, r or r + g; the two unsigned minima pick r.
__device__ __forceinline__ unsigned umod_tiny(unsigned a, unsigned g, float rg) {
  const unsigned q = (unsigned)((float)a * rg);
  unsigned r = a - __umul24(q, g);
  const unsigned up = r + g;
  r = up < r ? up : r;   // q one too large: a - q g wrapped around
  const unsigned dn = r - g;
  r = dn < r ? dn : r;   // q one too small
  return r;
}
// The same with a reciprocal scaled down by (1 - 2^-20), for 2 <= g < 2^20 and a < 2^20: rgl = v_rcp_f32(g) * (1 - 2^-20).
// With x = a / g < 2^19 the estimate a * rgl is x (1 - d), d in [0.75, 1.25] * 2^-20 (reciprocal 1 ulp, two roundings):
// below x, above x - 0.625 -- its floor is the quotient or one less, never more.  So a - q g is r or r + g and ONE unsigned
// minimum picks r (seven instructions per entry instead of nine).
__device__ __forceinline__ float rcp_low(unsigned g) { return __builtin_amdgcn_rcpf((float)g) * 0.99999904632568359375f; }
__device__ __forceinline__ unsigned umod_tiny_low(unsigned a, unsigned g, float rgl) {
  const unsigned q = (unsigned)((float)a * rgl);
  const unsigned r = a - __umul24(q, g);
  const unsigned dn = r - g;
  return dn < r ? dn : r;
}

// The same row update when every operand is small: the row's and the pivot row's entries below
// 2^15 (magnitude class 0) and |lpiv|, |foo|, |dpiv| < 2^15.  Then every product is below 2^30 and
// every z below 2^31: 24-bit multiplies (full rate, unlike the 64-bit product's three quarter-rate
// multiplies) and 32-bit registers all the way through the gcd refinement and the division give
// the same bits as the 64-bit code above.  64-bit entries only.
// Second half of the small row update: z (ints, |z| < 2^31, `mx` = OR of the lane's |z|) is divided by
// g = gcd(g0, z_0, ..., z_n) in place; newden = g0 / g.  False where the reference would divide by zero.
template <int NCH>
__device__ __forceinline__ bool small_reduce(int (&z)[NCH][2], unsigned mx, i64 g0, int lane, i64 &newden) {
  newden = g0;
  bool ok = true;
  if (g0 != 1) {
    u64 g64 = (u64)uni64((i64)uabs64(g0));
    if ((g64 >> 32) == 0) {
      unsigned g = (unsigned)g64;
      // every |z| and g below 2^20 (the rule on this path): remainders through a float reciprocal of the wave-uniform
      // g -- five full-rate instructions and two corrections per entry instead of the 32-bit division sequence
      // (g == 1 -- a denominator product of -1 -- takes the division path below: a % 1; inside the loop g >= 2)
      const bool tiny = g >= 2 && g < (1u << 20) && ballot64((mx >> 20) != 0) == 0;
      for (;;) {
        unsigned rr = 0;
        if (tiny) {
          const float rgl = rcp_low(g);
#pragma unroll
          for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const unsigned a = (unsigned)(z[c][h] < 0 ? -z[c][h] : z[c][h]);
              rr = rr ? rr : umod_tiny_low(a, g, rgl);
            }
        } else {
#pragma unroll
          for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const unsigned a = (unsigned)(z[c][h] < 0 ? -z[c][h] : z[c][h]);
              const unsigned m = g == 0 ? a : a % g;
              rr = rr ? rr : m;
            }
        }
        const u64 nz = ballot64(rr != 0);
        if (!nz) break;
        const unsigned r0 = __builtin_amdgcn_readlane(rr, __ffsll((long long)nz) - 1);
        g = gcd_u32(g, r0);
        if (g == 1) break;
      }
      if (g == 0) ok = false;  // the reference would divide by zero here
      if (g > 1) {
        const int sh = __builtin_ctz(g);
        const unsigned m = g >> sh;
        unsigned inv = m;  // 3 correct bits, doubled by every step
        inv *= 2u - m * inv;
        inv *= 2u - m * inv;
        inv *= 2u - m * inv;
        inv *= 2u - m * inv;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int v = z[c][h];
            const unsigned qq = ((unsigned)(v < 0 ? -v : v) >> sh) * inv;
            z[c][h] = v < 0 ? -(int)qq : (int)qq;
          }
        const unsigned qd = ((unsigned)uabs64(g0) >> sh) * inv;
        newden = g0 < 0 ? wneg((i64)qd) : (i64)qd;
      }
    } else if (ballot64(mx != 0) == 0) {
      // a zero row under a denominator beyond 32 bits: gcd(g0, 0, ..., 0) = |g0|
      newden = g0 < 0 ? -1 : 1;
    } else {
      // a denominator beyond 32 bits with 31-bit entries: the gcd is that of the entries (found by
      // the same refinement, starting from the first non-zero one) with the denominator -- gcd is
      // associative and commutative, so this is the reference's left fold
      unsigned first = 0;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const unsigned a = (unsigned)(z[c][h] < 0 ? -z[c][h] : z[c][h]);
          first = first ? first : a;
        }
      unsigned g = __builtin_amdgcn_readlane(first, __ffsll((long long)ballot64(first != 0)) - 1);
      while (g != 1) {
        unsigned rr = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const unsigned a = (unsigned)(z[c][h] < 0 ? -z[c][h] : z[c][h]);
            const unsigned m = a % g;
            rr = rr ? rr : m;
          }
        const u64 nz = ballot64(rr != 0);
        if (!nz) break;
        g = gcd_u32(g, __builtin_amdgcn_readlane(rr, __ffsll((long long)nz) - 1));
      }
      const unsigned g32 = (unsigned)gcd_mag(g64, (u64)g);  // divides the entries: fits 32 bits
      if (g32 > 1) {
        const int sh = __builtin_ctz(g32);
        const unsigned m = g32 >> sh;
        unsigned inv = m;
        inv *= 2u - m * inv;
        inv *= 2u - m * inv;
        inv *= 2u - m * inv;
        inv *= 2u - m * inv;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int v = z[c][h];
            const unsigned qq = ((unsigned)(v < 0 ? -v : v) >> sh) * inv;
            z[c][h] = v < 0 ? -(int)qq : (int)qq;
          }
        newden = cquo(g0, (i64)g32);
      }
    }
  }
  return ok;
}

template <int NCH>
__device__ __forceinline__ bool update_row_small(const RowRegs<i64, NCH> &r, RowRegs32<NCH> &out, const i64 *prow, int pivj,
                                                 int lp, int foo, int dpiv, i64 g0, int lane, i64 &newden) {
  int z[NCH][2];
  unsigned mx = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int j = colof<i64>(c, lane, h);
      const int p = (int)r.v[c][h], q = (int)prow[j];
      int v = __mul24(p, lp) - __mul24(q, foo);
      if (j == pivj) v = __mul24(dpiv, foo);
      z[c][h] = v;
      mx |= (unsigned)(v < 0 ? -v : v);
    }
  const bool ok = small_reduce<NCH>(z, mx, g0, lane, newden);
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < 2; h++) out.v[c][h] = z[c][h];
  return ok;
}

// integrer.c:98-150 bezout: z with z*y == x (mod delta) when gcd(y, delta) == 1, else 0.
// Wave-uniform scalar work (deepest-cut option only).
template <class T>
__device__ T bezout_dev(T x, T y, T delta) {
  T a = 1, b = 0, c = 0, d = 1, u = y, v = delta;
  for (int guard = 0; guard < 4 * ET<T>::BITS; guard++) {
    const T r = fmod64(u, v);
    const T q = cquo(wsub(u, r), v);  // floor division: (u - (u mod v)) / v is exact
    if (r == 0) break;
    u = v;
    v = r;
    const T e = wsub(a, wmul(q, c)), f = wsub(b, wmul(q, d));
    a = c;
    b = d;
    c = e;
    d = f;
  }
  if (v != 1) return 0;
  return fmod64(wmul(c, x), delta);
}

// flag exam_coef (traiter.c:121-154) gives an Unknown row, from its sign summary
__device__ __forceinline__ int exam_class(int sg) {
  const int fc = SIG_CONST(sg) == 1 ? PIPAMD_F_PLUS : (SIG_CONST(sg) == 2 ? PIPAMD_F_MINUS : PIPAMD_F_ZERO);
  const int pp = SIG_PPOS(sg), pn = SIG_PNEG(sg);
  if (pp && pn) return PIPAMD_F_UNKNOWN;
  if (pp) return (fc == PIPAMD_F_MINUS) ? PIPAMD_F_UNKNOWN : PIPAMD_F_PLUS;
  if (pn) return (fc != PIPAMD_F_MINUS) ? PIPAMD_F_UNKNOWN : PIPAMD_F_MINUS;
  return fc;
}

// ------------------------------------------------------------------ exam_coef
// traiter.c:101-159, general form (used when there is a big parameter).  Block-collective;
// returns the first row proven negative or BIG_I.  Rows are visited in logical order, which
// for the slot-indexed tables means "compare srow[s]".
template <class T, int NW>
__device__ int exam_rows(const Shared<T> &S, Scalars *sc, int ni) {
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x;
  if (tid == 0) sc->tmp = BIG_I;
  bsync<NW>();
  for (int s = tid; s < ni; s += NT)
    if (S.fl[s] == PIPAMD_F_UNKNOWN && SIG_BIG(S.sig[s]) == 2) atomicMin(&sc->tmp, (int)S.srow[s]);
  bsync<NW>();
  const int i1 = sc->tmp;
  for (int s = tid; s < ni; s += NT)
    if (S.fl[s] == PIPAMD_F_UNKNOWN) {
      const int k = S.srow[s];
      if (k == i1)
        S.fl[s] = PIPAMD_F_MINUS;
      else if (k < i1 && SIG_BIG(S.sig[s]) == 1)
        S.fl[s] = PIPAMD_F_PLUS;
    }
  bsync<NW>();
  if (i1 != BIG_I) return i1;
  if (tid == 0) sc->tmp = BIG_I;
  bsync<NW>();
  for (int s = tid; s < ni; s += NT) {
    int f = 0;
    if (S.fl[s] == PIPAMD_F_UNKNOWN) {
      f = exam_class(S.sig[s]);
      if (f == PIPAMD_F_MINUS) atomicMin(&sc->tmp, (int)S.srow[s]);
    }
    S.nf[s] = (u8)f;
  }
  bsync<NW>();
  const int i2 = sc->tmp;
  for (int s = tid; s < ni; s += NT)
    if (S.nf[s] && (int)S.srow[s] <= i2) S.fl[s] = S.nf[s];
  bsync<NW>();
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
//     cross-multiplication, ties kept); a real row that is zero in every tied
//     column cannot separate them, so it is skipped on its LDS bitmap alone;
// and stop when one column is left.  Executed by wave 0 only; `prow` holds the
// pivot row in the wave's lane geometry.
// Exact while (max a_j) * (max |entry|) < 2^62, which the caller guarantees.
// SMALL: every entry of every row (the pivot row included) is below 2^15 -- the caller knows from the magnitude
// classes -- so the candidates and the cross products are held and formed on 32-bit registers with 24-bit multiplies.
template <class T, int NCH, bool SMALL>
__device__ int choose_column(const Shared<T> &S, const RowRegs<T, NCH> &prow, const T *vals, int W, int nvar, int nligne,
                             int pivi, int ncolp, Scalars *sc) {
  constexpr int NM = NCH * ET<T>::CPL;
  const int lane = threadIdx.x & 63;
  typedef typename std::conditional<SMALL, int, T>::type A;  // arithmetic type of the ratio test
  A a[NCH][ET<T>::CPL];
  int u[NCH][ET<T>::CPL];
  bool cand[NCH][ET<T>::CPL];
  u64 cm[NM];
  int count = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < ET<T>::CPL; h++) {
      int j = colof<T>(c, lane, h);
      a[c][h] = j < nvar ? (A)prow.v[c][h] : (A)0;
      cand[c][h] = a[c][h] > 0;
      u[c][h] = cand[c][h] ? (int)S.urow[j] : -1;
      cm[ET<T>::CPL * c + h] = ballot64(cand[c][h]);
      count += __popcll(cm[ET<T>::CPL * c + h]);
    }
  if (count == 0) return -1;
  CNT(19, count == 1);
  for (int k0 = 0; k0 < nligne && count > 1; k0 += 64) {
    CNT(18, 1);
    const int k = k0 + lane;
    bool rel = false;
    if (k < nligne && k != pivi) {
      const int rf = S.ref[k];
      if (!(rf & UNITBIT)) {
        const u64 *m = S.nzm + (size_t)rf * NM;
        u64 x = 0;
#pragma unroll
        for (int e = 0; e < NM; e++) x |= m[e] & cm[e];
        rel = x != 0;
      }
    }
    u64 relmask = ballot64(rel);
    while (relmask && count > 1) {
      const int kk = k0 + __ffsll((long long)relmask) - 1;
      relmask &= relmask - 1;
      CNT(17, 1);
      const int sl = S.ref[kk];
      // unit rows above kk knock out their own column
      int nel = 0;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int h = 0; h < ET<T>::CPL; h++) nel += __popcll(ballot64(cand[c][h] && u[c][h] < kk));
      if (nel == count) goto last_unit_wins;
      if (nel) {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < ET<T>::CPL; h++) {
            if (u[c][h] < kk) cand[c][h] = false;
            cm[ET<T>::CPL * c + h] = ballot64(cand[c][h]);
          }
        count -= nel;
        if (count == 1) break;
      }
      {  // still able to separate the remaining columns?
        u64 x = 0;
#pragma unroll
        for (int e = 0; e < NM; e++) x |= S.nzm[(size_t)sl * NM + e] & cm[e];
        if (!x) continue;
      }
      // real row kk: keep the minimal ratios
      RowRegs<T, NCH> n;
      row_load<T, NCH>(n, vals + (size_t)sl * W, ncolp, lane);
      CNT(16, 1);
      for (;;) {
        CNT(15, 1);
        // reference column b = first remaining candidate
        A ab = 0, nb = 0;
        bool got = false;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < ET<T>::CPL; h++) {
            u64 m = cm[ET<T>::CPL * c + h];
            if (!got && m) {
              int src = __ffsll((long long)m) - 1;
              if constexpr (SMALL) {
                ab = __builtin_amdgcn_readlane(a[c][h], src);
                nb = __builtin_amdgcn_readlane((int)n.v[c][h], src);
              } else {
                ab = readlane64(a[c][h], src);
                nb = readlane64(n.v[c][h], src);
              }
              got = true;
            }
          }
        bool neg[NCH][ET<T>::CPL];
        int nneg = 0, nzero = 0;
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < ET<T>::CPL; h++) {
            A x;
            if constexpr (SMALL)
              x = __mul24(ab, (int)n.v[c][h]) - __mul24(nb, a[c][h]);
            else
              x = wsub(wmul(ab, n.v[c][h]), wmul(nb, a[c][h]));
            neg[c][h] = cand[c][h] && x < 0;
            bool zero = cand[c][h] && x == 0;
            nneg += __popcll(ballot64(neg[c][h]));
            nzero += __popcll(ballot64(zero));
            if (!neg[c][h] && !zero) cand[c][h] = false;  // strictly larger: out
          }
        if (nneg == 0) {
          count = nzero;
        } else {
#pragma unroll
          for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int h = 0; h < ET<T>::CPL; h++) cand[c][h] = neg[c][h];
          count = nneg;
        }
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
          for (int h = 0; h < ET<T>::CPL; h++) cm[ET<T>::CPL * c + h] = ballot64(cand[c][h]);
        if (nneg == 0 || count == 1) break;
      }
    }
  }
  if (count == 1) {
    int res = -1;
#pragma unroll
    for (int e = 0; e < NM; e++)
      if (cm[e]) res = colof<T>(e / ET<T>::CPL, __ffsll((long long)cm[e]) - 1, e % ET<T>::CPL);
    return res;
  }
last_unit_wins:
  // only unit rows left to look at: the column whose unit row comes last survives
  if (lane == 0) sc->tmp2 = -1;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < NCH; c++)
#pragma unroll
    for (int h = 0; h < ET<T>::CPL; h++)
      if (cand[c][h]) atomicMax(&sc->tmp2, (u[c][h] << 10) | colof<T>(c, lane, h));
  __builtin_amdgcn_wave_barrier();
  return sc->tmp2 & 1023;
}

// choisir_piv when entries are too large for the tournament's exactness guard: the reference's
// own fold (traiter.c:312-331), candidate by candidate, with its wrap-around products; only the
// search for the first row with a non-zero cross product is spread over the lanes.  Slow (two
// column gathers per candidate) but bit-identical whatever the magnitudes.  Wave 0.
template <class T>
__device__ int choose_column_slow(const Shared<T> &S, const T *vals, int W, int nvar, int nligne) {
  const int lane = threadIdx.x & 63;
  int pivj = -1;
  T pivot = 0;
  for (int j = 0; j < nvar; j++) {
    const T foo = S.prow[j];
    if (!(foo > 0)) continue;
    if (pivj < 0) {
      pivj = j;
      pivot = foo;
      continue;
    }
    bool less = false;
    for (int k0 = 0; k0 < nligne; k0 += 64) {
      const int k = k0 + lane;
      T x = 0;
      if (k < nligne) {
        const int rf = S.ref[k];
        T vj, vb;
        if (rf & UNITBIT) {  // valeur(): the unit row's denominator (1) in its own column
          vj = (UNITCOL(rf) == j) ? 1 : 0;
          vb = (UNITCOL(rf) == pivj) ? 1 : 0;
        } else {
          vj = vals[(size_t)rf * W + j];
          vb = vals[(size_t)rf * W + pivj];
        }
        x = wsub(wmul(pivot, vj), wmul(vb, foo));
      }
      const u64 nz = ballot64(x != 0);
      if (nz) {
        const int src = __ffsll((long long)nz) - 1;
        less = readlane64(x, src) < 0;
        break;
      }
    }
    if (less) {
      pivj = j;
      pivot = foo;
    }
  }
  return pivj;
}

// ------------------------------------------------------------ tab_sort_rows
// traiter.c:591-614: selection sort of the real rows nvar..nligne-1 by `size`
// (first minimum strictly below the running bound, swap into place).  With
// slot-indexed row data a swap of two logical rows is a swap of their slots.  Wave 0.
template <class T>
__device__ void sort_rows(const Shared<T> &S, int nvar, int nligne, double smax) {
  const int lane = threadIdx.x & 63;
  if (nligne - nvar <= 64) {
    // at most 64 candidate rows: lane l keeps logical row nvar+l (its slot, its key and whether
    // the selection may pick it) in registers.  Sizes are non-negative floats, so their bit
    // patterns order like the values.  Same selection and swaps as below.
    const int n = nligne - nvar;
    unsigned rf = lane < n ? S.ref[nvar + lane] : UNITBIT;
    const bool unit = (rf & UNITBIT) != 0;
    unsigned key = 0xFFFFFFFFu;  // rows the selection never picks (Unit, or size >= smax)
    if (!unit) {
      const float sj = S.size[rf];
      if ((double)sj < smax) key = __float_as_uint(sj);
    }
    const u64 units = ballot64(unit);
    for (int i = 0; i < n; i++) {
      if ((units >> i) & 1) continue;
      const unsigned m = wave_minmax_u32<false>(lane >= i ? key : 0xFFFFFFFFu);
      if (m == 0xFFFFFFFFu) continue;  // nothing below smax is left: row i stays
      const u64 hit = ballot64(lane >= i && key == m);
      const int pv = __builtin_ctzll(hit);
      if (pv != i) {
        const unsigned ki = __builtin_amdgcn_readlane(key, i), ri = __builtin_amdgcn_readlane(rf, i);
        const unsigned rp = __builtin_amdgcn_readlane(rf, pv);
        if (lane == pv) {
          key = ki;
          rf = ri;
        }
        if (lane == i) {
          key = m;
          rf = rp;
        }
      }
    }
    if (lane < n && !unit) S.ref[nvar + lane] = (u16)rf;
    __builtin_amdgcn_wave_barrier();
    return;
  }
  for (int i = nvar; i < nligne; i++) {
    if (S.ref[i] & UNITBIT) continue;
    float best = 0;
    int bj = BIG_I;
    for (int j = i + lane; j < nligne; j += 64) {
      const int rf = S.ref[j];
      if (rf & UNITBIT) continue;
      float sj = S.size[rf];
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
      u16 t = S.ref[pv];
      S.ref[pv] = S.ref[i];
      S.ref[i] = t;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// x86 cvttsd2si semantics of the reference's (int)t, traiter.c:583
__device__ __forceinline__ int trunc_int_x86(double t) {
  if (!(t > -2147483649.0 && t < 2147483648.0)) return (int)0x80000000;
  return (int)t;
}

// ================================================================ main kernel
// Work list of one launch (see pip_host.cpp, pipamd_batch_solve): workgroup b runs entry b of the
// input list.  The hardware's workgroup dispatcher is the queue: a workgroup ends when its job is
// finished, needs the host, has spent the launch's pivot budget or has no room left in the LDS
// image, and the next one starts in its place, so every CU stays busy while work is left.
//   in_list/in_count : jobs to run (NULL: all jobs 0..njobs-1); workgroups beyond *in_count exit
//   out_list/out_count/out_maxni : jobs still PIPAMD_ST_RUN when their workgroup let go of them and
//                      the largest row count among them -- the input of the next launch, without a
//                      host round trip in between (NULL: not recorded)
struct PipQueue {
  const int *in_list;
  const int *in_count;
  int *out_list;
  int *out_count;
  int *out_maxni;
};

// GM: the job's row tables (the "LDS image": Shared<T>) live in HBM instead of LDS -- one block of
// `gimg_bytes` per workgroup at `gimg` -- for jobs whose tables outgrow the 159 KiB a workgroup can
// get.  Same code, every table access becomes a global access (cached in this CU's L1/L2); only
// the handful of workgroup scalars stay in LDS.
// SC > 0: the row capacity of the LDS image is the compile-time constant SC (and SC + WP logical
// rows) instead of the launch parameters Smax / Lmax.  Every table of the image then sits at a
// constant LDS address: the thirteen base pointers need no scalar registers and no address
// arithmetic per access (the offsets fold into the ds_ instructions).  Used for the bulk launches
// of the common shapes (launch_advance_w picks the smallest class that holds the launch).
// FULL: every job of the launch has no parameters, no big parameter and rows that fill the wave's
// registers exactly (nvar + 1 == W == the columns a wave covers, e.g. 127 unknowns + constant):
// column counts and the row stride are compile-time constants, so the per-column range checks, the
// parameter-sign bookkeeping of the row summaries and the stride multiplications disappear.
template <class T, int NCH, int NW, bool GM, int SC, bool FULL>
__global__ __launch_bounds__(64 * NW, (NW == 1 && NCH == 1 && sizeof(T) == 8) ? PIP_MINWAVES
                                      : ((sizeof(T) == 16 && NCH <= 4) ? (NW == 16 ? 4 : PIP_MINWAVES128) : 1)) void pip_advance_kernel(
    PipJob *jobs, i64 *arena, int njobs, int Lmax_, int Smax_, int Wmax, int iter_limit, PipQueue q, unsigned char *gimg,
    size_t gimg_bytes, int gslots, u64 *prof) {
  const int Smax = SC > 0 ? SC : Smax_;
  const int Lmax = SC > 0 ? SC + NCH * 64 * ET<T>::CPL : Lmax_;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ Scalars sc;
  (void)Wmax;
  const int nq = q.in_count ? *q.in_count : njobs;
  if ((int)blockIdx.x >= nq) return;
  const int jb = q.in_list ? q.in_list[blockIdx.x] : (int)blockIdx.x;
  PipJob *J = &jobs[jb];
  if (J->status != PIPAMD_ST_RUN) {
    // out of spare rows in an earlier launch: stays on the list until the host has re-housed it (pip_rehouse_kernel)
    if (J->status == PIPAMD_ST_CAPACITY && q.out_count && threadIdx.x == 0) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, PIPAMD_Q_CAPFLAG | J->ni);
      atomicAdd(q.out_maxni + 1, 1);  // (the list\'s third control word: tableaux out of rows)
    }
    return;
  }
  constexpr int NT = 64 * NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int WP = NCH * 64 * ET<T>::CPL;  // columns a wave's registers cover; prow/urow are padded to it
  constexpr int NM = NCH * ET<T>::CPL;
  PROF_DECL;
#ifdef PIP_PROFILE_EVENTS
  if (threadIdx.x == 0) pf_buf = prof;
#endif

  if constexpr (FULL) {
    if (J->nvar != WP - 1 || J->nparm != 0 || J->bigparm >= 0 || J->W != WP) {  // the launcher's promise does not hold
      if (tid == 0) J->status = PIPAMD_ST_INTERNAL;
      return;
    }
  }
  const int nvar = FULL ? WP - 1 : J->nvar, nparm = FULL ? 0 : J->nparm, bigparm = FULL ? -1 : J->bigparm;
  const bool has_parm = nparm > 0;
  int tflags = J->tflags;
  int ni = J->ni;
  const int L = J->L, Sl = J->S, W = FULL ? WP : J->W;
  const int ncol = nvar + nparm + 1;
  const int ncolp = ET<T>::CPL == 2 ? ((ncol + 1) & ~1) : ncol;  // rows are whole 16-byte units
  T *vals = (T *)(arena + J->vals_off);
  T *g_den = (T *)(arena + J->rows_off);
  int *g_flag = (int *)(g_den + L);
  int *g_ref = g_flag + L;
  int nligne = nvar + ni;
  int npiv = J->npiv, ncut = J->ncut, nupd = J->nupd;
  // The determinant limbs (traiter.c:412-446) are not updated here: every pivot appends (pivot,
  // denominator of the pivot row) to the job's log, the pip_det_replay kernels run the bookkeeping
  // and its "Integer overflow" tests after the launch
  T *g_log = (T *)(arena + J->log_off);
  constexpr int LOGCAP = PIPAMD_DETLOG;  // pairs the log area holds
  int nlog = J->nlog;
  if (ni > Smax || nligne > Lmax) {  // this launch's LDS image is too small: stay RUN for a larger one
    if (tid == 0 && q.out_count) {
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, ni);
    }
    return;
  }
  Shared<T> S;
  int gslot = 0;
  (void)gslot;
  {
    unsigned char *p = smem;
    if constexpr (GM) {
      // the image lives in one of `gslots` HBM blocks behind an array of lock words: workgroup b takes block
      // b % gslots and waits for an earlier holder to leave (a holder is resident and running, so it does)
      gslot = (int)(blockIdx.x % (unsigned)gslots);
      if (tid == 0)
        while (atomicCAS((int *)gimg + gslot, 0, 1) != 0) __builtin_amdgcn_s_sleep(32);
      __syncthreads();
      p = gimg + (((size_t)gslots * sizeof(int) + 255) & ~(size_t)255) + (size_t)gslot * gimg_bytes;
    }
    S.den = (T *)p;      p += sizeof(T) * Smax;
    // the sort keys are dead once the rows are sorted (before the first pivot row is staged):
    // they share prow's storage, which is sized for the larger of the two
    S.prow = (T *)p;
    S.size = (float *)p;
    p += prow_bytes(sizeof(T) * WP, Smax);
    S.cst = (T *)p;      p += sizeof(T) * Smax;
    S.nzm = (u64 *)p;    p += sizeof(u64) * (size_t)Smax * NM;
    S.sig = (u16 *)p;    p += sizeof(u16) * Smax;
    S.srow = (u16 *)p;   p += sizeof(u16) * Smax;
    S.work = (u16 *)p;   p += sizeof(u16) * Smax;
    S.ref = (u16 *)p;    p += sizeof(u16) * Lmax;
    S.urow = (u16 *)p;   p += sizeof(u16) * WP;
    S.fl = (u8 *)p;      p += Smax;
    S.nf = (u8 *)p;      p += Smax;
    S.rcls = (u8 *)p;    p += Smax;
  }

  // saved LDS state of a paused job (bitmaps, sign summaries, magnitude classes)
  u64 *g_nzm = (u64 *)(arena + J->state_off);
  u16 *g_sig = (u16 *)(g_nzm + (size_t)Sl * NM);
  u8 *g_rcls = (u8 *)(g_sig + Sl);

  // A job loaded with PIPAMD_T_ROWS_STAY: its rows are still in the caller's array (slot s = input row s, pitch
  // ncol).  The one-wave bulk kernels fetch them in the pass that builds the summaries (FUSE, below); the other
  // instantiations copy them into the block first, a row per wave at a time, and read them back from L2 -- folded
  // into their summary pass, the extra pointer cost the four-wave kernel 6 % with row skipping off.
  constexpr bool FUSE = NW == 1 && SC > 0 && ET<T>::EW == 1;
  if constexpr (ET<T>::EW == 1 && !FUSE) {
    if (tflags & PIPAMD_T_FRESHROWS) {
      const T *fresh = (const T *)(uintptr_t)J->src_rows;
      for (int s = wave; s < ni; s += NW) {
        RowRegs<T, NCH> r;
        row_load<T, NCH>(r, fresh + (size_t)s * ncol, ncolp, lane);
        row_store<T, NCH>(r, vals + (size_t)s * W, ncolp, lane);
      }
      tflags &= ~PIPAMD_T_FRESHROWS;
      __threadfence_block();
      bsync<NW>();
    }
  }
  // ---- stage the row tables in LDS -------------------------------------
  for (int j = tid; j < WP; j += NT) S.urow[j] = NOROW;  // prow is written whole by every phase A
  if (tid == 0) {
    sc.ovf = 0;
    sc.aux = 0;
    sc.smaxbits = 0;
    sc.pivi = BIG_I;
    sc.pivi2 = BIG_I;
    sc.flagor = 0;
    sc.bad = 0;
  }
  bsync<NW>();
  for (int i = tid; i < nligne; i += NT) {
    const int f = g_flag[i], rf = g_ref[i];
    if (f & PIPAMD_F_UNIT) {
      S.ref[i] = (u16)(UNITBIT | ((f & PIPAMD_F_ZERO) ? UNITZERO : 0) | rf);
      S.urow[rf] = (u16)i;
    } else {
      S.ref[i] = (u16)rf;
      S.srow[rf] = (u16)i;
      S.fl[rf] = (u8)f;
      S.den[rf] = g_den[i];
      S.size[rf] = 0.f;
      S.nf[rf] = 0;
    }
  }
  bsync<NW>();
  PROF(13);
  if ((tflags & PIPAMD_T_STATE) && J->state_nch == NCH) {
    // resumed job: the summaries were saved when it paused
    for (int s = tid; s < ni; s += NT) {
      S.sig[s] = g_sig[s];
      S.rcls[s] = g_rcls[s];
    }
    for (int e = tid; e < ni * NM; e += NT) S.nzm[e] = g_nzm[e];
    for (int s = tid; s < ni; s += NT) S.cst[s] = vals[(size_t)s * W + nvar];
  } else {
    // one pass over the tableau: sign summaries, bitmaps, magnitudes, sort keys (PF rows of a
    // wave in flight at a time)
    constexpr int PF0 = (NCH * ET<T>::EW) <= 2 ? 4 : ((NCH * ET<T>::EW) <= 4 ? 2 : 1);  // by the registers a row takes
    const T *fresh = nullptr;
    if constexpr (FUSE)
      if (tflags & PIPAMD_T_FRESHROWS) fresh = (const T *)(uintptr_t)J->src_rows;
    for (int s0 = wave; s0 < ni; s0 += NW * PF0) {
      RowRegs<T, NCH> rr[PF0];
#pragma unroll
      for (int q = 0; q < PF0; q++)
        if (s0 + q * NW < ni) {
          const int s = s0 + q * NW;
          if (FUSE && fresh) {
            row_load<T, NCH>(rr[q], fresh + (size_t)s * ncol, ncolp, lane);
            row_store<T, NCH>(rr[q], vals + (size_t)s * W, ncolp, lane);
          } else {
            row_load<T, NCH>(rr[q], vals + (size_t)s * W, ncolp, lane);
          }
        }
#pragma unroll
      for (int q = 0; q < PF0; q++) {
        const int s = s0 + q * NW;
        if (s >= ni) break;
        RowRegs<T, NCH> &r = rr[q];
        // rows with a denominator other than 1 are conservatively treated as not yet reduced
        const bool den1 = S.den[s] == 1;
        row_publish<T, NCH>(r, S, s, nvar, ncol, bigparm, -1, den1 ? SIG_RED : 0, has_parm, lane);
        if (tflags & PIPAMD_T_SORT) {
          // traiter.c:576-589: size = max_j |(int)(v_j / den)| over the unknowns.  The per-entry
          // terms are ints (x86 cvttsd2si: INT_MIN when out of range, and abs(INT_MIN) stays
          // negative, so it never wins the max): the row maximum is in [0, 2^31).
          int sz = 0;
          if (den1) {
            // x / 1.0 == x exactly, and (int)x is x itself when it fits an int
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
              for (int h = 0; h < ET<T>::CPL; h++) {
                int j = colof<T>(c, lane, h);
                if (j < nvar) {
                  const T v = r.v[c][h];
                  const int q2 = (v == (T)(int)v) ? (int)v : (int)0x80000000;
                  const int aq = q2 < 0 ? (int)(0u - (unsigned)q2) : q2;
                  sz = sz > aq ? sz : aq;
                }
              }
          } else {
            const double d = to_double(S.den[s]);
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
              for (int h = 0; h < ET<T>::CPL; h++) {
                int j = colof<T>(c, lane, h);
                if (j < nvar) {
                  const int q2 = trunc_int_x86(to_double(r.v[c][h]) / d);
                  const int aq = q2 < 0 ? (int)(0u - (unsigned)q2) : q2;
                  sz = sz > aq ? sz : aq;
                }
              }
          }
          const unsigned szw = wave_minmax_u32<true>((unsigned)sz);
          if (lane == 0) {
            S.size[s] = (float)szw;
            // smax is taken over rows nvar..nligne-1 only (traiter.c:576-586)
            if ((int)S.srow[s] >= nvar) atomicMax(&sc.smaxbits, (u64)szw);
          }
        }
      }
    }
  }
  if constexpr (FUSE) tflags &= ~PIPAMD_T_FRESHROWS;
  bsync<NW>();
  PROF(14);
  if (tflags & PIPAMD_T_SORT) {
    if (wave == 0) sort_rows(S, nvar, nligne, (double)sc.smaxbits);
    bsync<NW>();
    for (int i = tid; i < nligne; i += NT)
      if (!(S.ref[i] & UNITBIT)) S.srow[S.ref[i]] = (u16)i;
    tflags &= ~PIPAMD_T_SORT;
    bsync<NW>();
  }
  PROF(15);
  // chercher(Minus) and the tentative exam_coef flags for the first iteration; later
  // iterations get both from phase C
  for (int s = tid; s < ni; s += NT) {
    const int ff = S.fl[s];
    if (ff & PIPAMD_F_MINUS)
      atomicMin(&sc.pivi, (int)S.srow[s]);
    else if (ff == PIPAMD_F_UNKNOWN && bigparm < 0) {
      const int ec = exam_class(S.sig[s]);
      S.nf[s] = (u8)ec;
      if (ec == PIPAMD_F_MINUS) atomicMin(&sc.pivi2, (int)S.srow[s]);
    }
  }
  bsync<NW>();

  int status = PIPAMD_ST_RUN;
  PROF(0);
  for (int iter = 0;; iter++) {
    if (iter >= iter_limit) break;  // status stays RUN: the host relaunches
    if (nlog >= LOGCAP) break;  // determinant log full: likewise
    int pivi = sc.pivi;
    if (pivi == BIG_I) {
      // -------------- exam_coef, then (if nothing is negative) integrer ---------
      if (bigparm >= 0) {
        pivi = exam_rows<T, NW>(S, &sc, ni);
      } else {
        // the flags exam_coef would assign were computed with the post-pivot hints; they are
        // applied up to the first row it proves negative (traiter.c:154-156)
        for (int rep11 = 0; rep11 < PIP_DUP_REPS(11); rep11++) {
          if (PIP_DUP == 11) PIP_OPAQUE_MEM();
          pivi = sc.pivi2;
          for (int s = tid; s < ni; s += NT)
            if (S.fl[s] == PIPAMD_F_UNKNOWN && (int)S.srow[s] <= pivi) S.fl[s] = S.nf[s];
          bsync<NW>();
        }
      }
      PROF(1);
      if (pivi == BIG_I) {
        if (has_parm) {
          for (int s = tid; s < ni; s += NT)
            if (S.fl[s] & (PIPAMD_F_CRITIC | PIPAMD_F_UNKNOWN)) atomicOr(&sc.flagor, 1);
          bsync<NW>();
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
        bsync<NW>();
        for (int rep19 = 0; rep19 < PIP_DUP_REPS(19); rep19++)
        for (int i = tid; i < nvar; i += NT) {
          if (PIP_DUP == 19) PIP_OPAQUE_MEM();
          const int rf = S.ref[i];
          if (rf & UNITBIT) continue;
          const T D = S.den[rf];
          if (D == 1) continue;
          bool ok = wneg(fmod64(wneg(S.cst[rf]), D)) != 0;
          if (has_parm && !ok) {
            const T *row = vals + (size_t)rf * W;
            for (int j = nvar + 1; j < ncol && !ok; j++)
              if (j != bigparm && fmod64(wneg(row[j]), D) != 0) ok = true;
          }
          if (ok) atomicMin(&sc.tmp, i);
        }
        bsync<NW>();
        const int ci = sc.tmp;
        if (ci == BIG_I) {
          status = PIPAMD_ST_SOLUTION;
          break;
        }
        // wave 0 builds the cut in its registers (integrer.c:357-386) and appends it
        if (wave == 0) {
          const int cslot = S.ref[ci];
          const T D = uni64(S.den[cslot]);
          RowRegs<T, NCH> r;
          bool okv = false, okp = false;
          for (int rep20 = 0; rep20 < PIP_DUP_REPS(20); rep20++) {
          if (PIP_DUP == 20) PIP_OPAQUE_MEM();
          row_load<T, NCH>(r, vals + (size_t)cslot * W, ncolp, lane);
#pragma unroll
          for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int h = 0; h < ET<T>::CPL; h++) {
              int j = colof<T>(c, lane, h);
              T v = r.v[c][h], x = 0;
              if (j < nvar) {
                x = fmod64(v, D);
                okv |= x > 0;
              } else if (j == nvar) {
                x = wneg(fmod64(wneg(v), D));
              } else if (j < ncol && j != bigparm) {
                x = wneg(fmod64(wneg(v), D));
                okp |= x != 0;
              }
              r.v[c][h] = x;
            }
          }
          const bool any_v = ballot64(okv) != 0, any_p = ballot64(okp) != 0;
          int verdict;
          if (any_p)
            verdict = PIPAMD_ST_NEED_PARMCUT;  // the host owns the context (find_parm/add_parm)
          else if (!any_v)
            verdict = PIPAMD_ST_NIL;  // integrer.c:482-485 case (b)
          else if (ni >= Sl || nligne >= L)
            verdict = PIPAMD_ST_CAPACITY;
          else if (ni >= Smax || nligne >= Lmax)
            verdict = -1;  // no room in this launch's LDS image: pause, the host relaunches with more
          else {
            verdict = PIPAMD_ST_RUN;
            if (tflags & PIPAMD_T_DEEPEST) {
              // deepest cut, integrer.c:417-438: scale the cut by the multiplier lambda that
              // makes its constant term -1/D-tight
              constexpr int CW = 64 * ET<T>::CPL;
              const T cn = row_entry<T, NCH>(r, nvar / CW, nvar % ET<T>::CPL, (nvar % CW) / ET<T>::CPL);
              T t = wneg(cn);
              const T delta = gcd_i64(t, D), tau = cquo(t, delta), dd = cquo(D, delta);
              t = wsub(dd, (T)1);
              T lambda = bezout_dev<T>(t, tau, dd);
              t = gcd_i64(lambda, D);
              for (int guard = 0; t != 1 && guard < (1 << 20); guard++) {
                lambda = wadd(lambda, dd);
                t = gcd_i64(lambda, D);
              }
#pragma unroll
              for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int h = 0; h < ET<T>::CPL; h++) {
                  int j = colof<T>(c, lane, h);
                  if (j < nvar)
                    r.v[c][h] = fmod64(wmul(lambda, r.v[c][h]), D);
                  else if (j == nvar)
                    r.v[c][h] = wneg(wsub(D, fmod64(wmul(r.v[c][h], lambda), D)));
                }
            }
            // append the cut as logical row nligne in slot ni (integrer.c:440-446)
            for (int rep21 = 0; rep21 < PIP_DUP_REPS(21); rep21++) {
              if (PIP_DUP == 21) PIP_OPAQUE_MEM();
              row_store<T, NCH>(r, vals + (size_t)ni * W, ncolp, lane);
              row_publish<T, NCH>(r, S, ni, nvar, ncol, bigparm, -1, 0, has_parm, lane);
            }
            if (lane == 0) {
              S.fl[ni] = PIPAMD_F_MINUS;
              S.nf[ni] = 0;
              S.den[ni] = D;
              S.ref[nligne] = (u16)ni;
              S.srow[ni] = (u16)nligne;
            }
          }
          if (lane == 0) {
            sc.tmp2 = verdict;
            sc.aux = ci;
          }
        }
        bsync<NW>();
        if (sc.tmp2 != PIPAMD_ST_RUN) {
          status = sc.tmp2 < 0 ? PIPAMD_ST_RUN : sc.tmp2;
          break;
        }
        pivi = nligne;
        ni++;
        nligne++;
        ncut++;
      }
      PROF(2);
    }
    // ---------------- A (wave 0): pivot row, choisir_piv, work list ------------
    npiv++;
    const int pslot = S.ref[pivi];
    if (wave == 0) {
      RowRegs<T, NCH> pr;
      int mc = 0;
      typename ET<T>::U amax = 0;
      for (int rep14 = 0; rep14 < PIP_DUP_REPS(14); rep14++) {
      if (PIP_DUP == 14) PIP_OPAQUE_MEM();
      row_load<T, NCH>(pr, vals + (size_t)pslot * W, ncolp, lane);
      // (while the pivot row is on its way) largest magnitude class of any row, for the guard below
      mc = 0;
      for (int s = lane; s < ni; s += 64)
        if (S.rcls[s] > mc) mc = S.rcls[s];
      mc = ballot64(mc == 3) ? 3 : (ballot64(mc == 2) ? 2 : (ballot64(mc == 1) ? 1 : 0));
      amax = 0;
#pragma unroll
      for (int c = 0; c < NCH; c++)
#pragma unroll
        for (int h = 0; h < ET<T>::CPL; h++) {
          int j = colof<T>(c, lane, h);
          if (j >= ncol) pr.v[c][h] = 0;
          S.prow[j] = pr.v[c][h];
          if (j < nvar && pr.v[c][h] > 0) amax |= (typename ET<T>::U)pr.v[c][h];
        }
      }
      // exactness guard of the tournament: (max candidate a_j) * (max |entry|) < 2^62
      const int abits = cls_bits<T>(cls_of<T>(amax));
      const bool safe = abits + cls_bits<T>(mc) <= ET<T>::BITS - 2;
      PROF(3);
      int pj;
      if (PIP_OPT_CSMALL && sizeof(T) == 8 && mc == 0)  // every row in class 0, the pivot row among them
        pj = choose_column<T, NCH, sizeof(T) == 8>(S, pr, vals, W, nvar, nligne, pivi, ncolp, &sc);
      else
        pj = safe ? choose_column<T, NCH, false>(S, pr, vals, W, nvar, nligne, pivi, ncolp, &sc)
                  : choose_column_slow(S, vals, W, nvar, nligne);
      if (PIP_DUP == 15) {
        PIP_OPAQUE_MEM();
        asm volatile("" : "+v"(pr.v[0][0]));
        pj = safe ? choose_column<T, NCH, false>(S, pr, vals, W, nvar, nligne, pivi, ncolp, &sc)
                  : choose_column_slow(S, vals, W, nvar, nligne);
      }
      PROF(4);
      if (pj >= 0) {
        // slots the elimination has to rewrite: the recycled pivot slot plus every real row
        // that is non-zero in column pj or not yet reduced
        constexpr int CW = 64 * ET<T>::CPL;
        const int pe = (pj / CW) * ET<T>::CPL + (pj % ET<T>::CPL), pl = (pj % CW) / ET<T>::CPL;
        int base = 0;
        for (int rep13 = 0; rep13 < PIP_DUP_REPS(13); rep13++) {
        if (PIP_DUP == 13) PIP_OPAQUE_MEM();
        base = 0;
        for (int s0 = 0; s0 < ni; s0 += 64) {
          const int s = s0 + lane;
          bool need = false;
          if (s < ni) {
            if (s == pslot)
              need = true;
            else {
              const bool nzb = (S.nzm[(size_t)s * NM + pe] >> pl) & 1;
              if (nzb || !(S.sig[s] & SIG_RED) || (tflags & PIPAMD_T_NOSKIP))
                need = true;
              else
                S.sig[s] &= ~0xC0;  // entry in the pivot column is 0: sign hint "zero"
            }
          }
          const u64 m = ballot64(need);
          if (need) S.work[base + __popcll(m & ((1ull << lane) - 1))] = (u16)s;
          base += __popcll(m);
        }
        }
        if (lane == 0) sc.nwork = base;
      }
      if (lane == 0) sc.pivj = pj;
    }
    bsync<NW>();
    if (tid == 0) {  // every wave has read them by now; phase C refills them
      sc.pivi = BIG_I;
      sc.pivi2 = BIG_I;
    }
    // everything phase A left in LDS for this point is read in one go (one wait instead of a
    // chain of round trips)
    const int pivj = sc.pivj, nwork = sc.nwork;
    const int wq0 = S.work[wave], wq1 = S.work[wave + NW < Smax ? wave + NW : 0];
    const T dpiv_v = S.den[pslot];
    const int psig_v = S.sig[pslot];
    const int prow_cls = S.rcls[pslot];  // magnitude class of the pivot row (as last published)
    if (pivj == -1) {  // traiter.c:782-785
      status = PIPAMD_ST_NIL;
      break;
    }
    if (pivj == -2) {
      status = PIPAMD_ST_RANGE;
      break;
    }
    // The first rows of the work list are requested from HBM before the scalar bookkeeping
    // below, so that their latency overlaps it (up to PF rows per wave in flight: a wave owns
    // only a few rows per pivot on sparse tableaux).
#ifndef PIP_PF
#define PIP_PF 2
#endif
    constexpr int PF = (NCH * ET<T>::EW) <= 2 ? PIP_PF : ((NCH * ET<T>::EW) <= 4 ? 2 : 1);  // by the registers a row takes
    RowRegs<T, NCH> rr[PF];
    if constexpr (!(sizeof(T) == 16 && PIP_OPT_LANEPREP)) {  // (the 128-bit row loop prepares its rows first, see below)
#pragma unroll
      for (int q = 0; q < PF; q++) {
        const int w = wave + q * NW;
        const int sw = q == 0 ? wq0 : (q == 1 ? wq1 : (int)S.work[w < nwork ? w : 0]);
        if (w < nwork && sw != pslot) row_load<T, NCH>(rr[q], vals + (size_t)sw * W, ncolp, lane);
      }
    }
    // pivot scalars, traiter.c:394-396 (uniform, every thread); the determinant bookkeeping of
    // traiter.c:412-446 only needs them logged
    const T pivot = uni64(S.prow[pivj]);
    const T dpiv = uni64(dpiv_v);
    if (tid == 0) {
      g_log[2 * nlog] = pivot;
      g_log[2 * nlog + 1] = dpiv;
    }
    nlog++;
    const int ku = S.urow[pivj];  // unit row of the entering column
    const int pred = psig_v & SIG_RED;
    const int pc = pivj / (64 * ET<T>::CPL), ph = pivj % ET<T>::CPL, pl = (pivj % (64 * ET<T>::CPL)) / ET<T>::CPL;
    PROF(5);
    // ---------------- B: eliminate the pivot column (all waves) ----------------
    if constexpr (sizeof(T) == 16 && PIP_OPT_LANEPREP) {
      // 128-bit entries: the wave-uniform part of a row's update -- gcd(pivot, foo), the multipliers, and the gcd of the
      // new denominator lp * den with the one entry of the new row known beforehand, dpiv * foo (row_reduce_wide's first
      // fold) -- is a couple of 128-bit binary gcds, thousands of instructions a row when every lane computes the same one,
      // several times the row's own arithmetic.  So a wave first prepares ALL its rows of this pivot, one row per lane
      // (the row's entry in the pivot column is gathered from HBM, its denominator comes from LDS), and the row loop
      // picks each row's scalars out of the lanes.
      nupd += nwork - 1;
      for (int wb = wave; wb < nwork; wb += NW * 64) {
        T m_lp = pivot, m_foo = 0;
        typename ET<T>::U m_g = 0;
        {
          const int wk = wb + lane * NW;
          const int sk = wk < nwork ? (int)S.work[wk] : pslot;
          if (sk != pslot) {
            T fk = vals[(size_t)sk * W + pivj];
            const T dk = S.den[sk];
            T lpk = pivot, g0k = dk;
            if (pivot != 1) {
              const T d = gcd_i64(pivot, fk);
              if (d != 1) {
                lpk = exact_quo(pivot, d);
                fk = exact_quo(fk, d);
              }
              g0k = wmul(lpk, dk);
            }
            m_lp = lpk;
            m_foo = fk;
            const T zf = wmul(dpiv, fk);
            typename ET<T>::U g = uabs64(g0k);
            if (g > 1 && zf != 0) g = gcd_mag(g, uabs64(zf));
            m_g = zf != 0 ? g : 0;  // (0: nothing folded in, row_reduce_wide starts from |g0|)
          }
        }
        for (int k = 0; k < 64; k++) {
          const int w = wb + k * NW;
          if (w >= nwork) break;
          const int s = S.work[w];
          T *row = vals + (size_t)s * W;
          RowRegs<T, NCH> r;
          if (s == pslot) {
            // the slot is recycled for the row replacing ku's unit row (traiter.c:461-465,503-513)
#pragma unroll
            for (int c = 0; c < NCH; c++) {
              const int j = colof<T>(c, lane, 0);
              r.v[c][0] = (j == pivj) ? dpiv : wneg(S.prow[j]);
            }
            row_store<T, NCH>(r, row, ncolp, lane);
            row_publish<T, NCH>(r, S, s, nvar, ncol, bigparm, pivj, pred, has_parm, lane);
            continue;
          }
          row_load<T, NCH>(r, row, ncolp, lane);
          const T lp = readlane64(m_lp, k), foo = readlane64(m_foo, k);
          const typename ET<T>::U gpre = (typename ET<T>::U)readlane64((T)m_g, k);
          PROF(9);
          if (foo == 0 && (S.sig[s] & SIG_RED)) {
            // only reached with PIPAMD_T_NOSKIP: multipliers (1, 0) and gcd 1, the reference rewrites the row with the
            // bits it read -- so do we, without the arithmetic
            row_store<T, NCH>(r, row, ncolp, lane);
            if (lane == 0) S.sig[s] &= ~0xC0;
            continue;
          }
          const T den_s = uni64(S.den[s]);
          const T g0 = pivot != 1 ? wmul(lp, den_s) : den_s;
          PROF(10);
          T nd;
          bool done_small = false;
          {
            // operands all below 2^31 (both rows in magnitude class 0) under a denominator product of at most 63 bits:
            // the update on 64-bit registers (update_row_narrow)
            const T lim = (T)1 << 31, glim = (T)1 << 62;
            if (PIP_OPT_NARROW128 && S.rcls[s] == 0 && prow_cls == 0 && lp < lim && foo < lim && foo > -lim && dpiv < lim &&
                dpiv > -lim && g0 < glim && g0 > -glim) {
              if (!update_row_narrow<NCH>(r, S.prow, pivj, (i64)lp, (i64)foo, (i64)dpiv, (i64)g0, lane, nd, (u64)gpre)) {
                if (lane == 0) sc.bad = 1;
              }
              done_small = true;
            }
          }
          // (classes of the 128-bit kernel: 0 below 2^31, 1 below 2^63)
          const bool n64 = S.rcls[s] <= 1 && prow_cls <= 1 && fits64(lp) && fits64(foo);
          if (!done_small && !update_row<T, NCH>(r, S.prow, pivj, lp, foo, dpiv, g0, lane, nd, gpre, n64)) {
            if (lane == 0) sc.bad = 1;
          }
          PROF(11);
          row_store<T, NCH>(r, row, ncolp, lane);
          row_publish<T, NCH>(r, S, s, nvar, ncol, bigparm, pivj, SIG_RED, has_parm, lane);
          if (lane == 0) S.den[s] = nd;
          PROF(12);
        }
      }
    } else {
      nupd += nwork - 1;
      for (int w0 = wave; w0 < nwork; w0 += NW * PF) {
        if (w0 != wave) {  // the first PF rows are already on their way
#pragma unroll
          for (int q = 0; q < PF; q++) {
            const int w = w0 + q * NW;
            if (w < nwork && S.work[w] != pslot) row_load<T, NCH>(rr[q], vals + (size_t)S.work[w] * W, ncolp, lane);
          }
        }
#pragma unroll
        for (int q = 0; q < PF; q++) {
          const int w = w0 + q * NW;
          if (w >= nwork) break;
          RowRegs<T, NCH> &r = rr[q];
          const int s = S.work[w];
          T *row = vals + (size_t)s * W;
          if (s == pslot) {
            // the slot is recycled for the row replacing ku's unit row (traiter.c:461-465,503-513)
            bool recycled32 = false;
            if constexpr (sizeof(T) == 8) {
              const T lim_ = (T)1 << 15;
              if (prow_cls == 0 && dpiv < lim_ && dpiv > -lim_ && !has_parm) {
                RowRegs32<NCH> z32;
#pragma unroll
                for (int c = 0; c < NCH; c++)
#pragma unroll
                  for (int h = 0; h < 2; h++) {
                    const int j = colof<T>(c, lane, h);
                    z32.v[c][h] = (j == pivj) ? (int)dpiv : -(int)S.prow[j];
                  }
                row_store32<NCH>(z32, row, ncolp, lane);
                row_publish32<NCH>(z32, S, s, nvar, pivj, pred, lane);
                recycled32 = true;
              }
            }
            if (!recycled32) {
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
              for (int h = 0; h < ET<T>::CPL; h++) {
                int j = colof<T>(c, lane, h);
                r.v[c][h] = (j == pivj) ? dpiv : wneg(S.prow[j]);
              }
            row_store<T, NCH>(r, row, ncolp, lane);
            row_publish<T, NCH>(r, S, s, nvar, ncol, bigparm, pivj, pred, has_parm, lane);
            }
            if (PIP_DUP == 9) {  // the recycled row once more
              PIP_OPAQUE_MEM();
#pragma unroll
              for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int h = 0; h < ET<T>::CPL; h++) {
                  int j = colof<T>(c, lane, h);
                  r.v[c][h] = (j == pivj) ? dpiv : wneg(S.prow[j]);
                }
              row_store<T, NCH>(r, row, ncolp, lane);
              row_publish<T, NCH>(r, S, s, nvar, ncol, bigparm, pivj, pred, has_parm, lane);
            }
          } else {
            T nd;
            if (PIP_DUP == 12) {  // the row's load and pivot-column read once more
              PIP_OPAQUE_MEM();
              row_load<T, NCH>(r, vals + (size_t)S.work[w] * W, ncolp, lane);
            }
            // multipliers from the row's own pivot-column entry (traiter.c:470-476)
            T foo = row_entry<T, NCH>(r, pc, ph, pl);
            if (PIP_DUP == 12) {
              asm volatile("" : "+v"(r.v[0][0]));
              foo ^= row_entry<T, NCH>(r, pc, ph, pl);
              foo = row_entry<T, NCH>(r, pc, ph, pl);
            }
            PROF(9);
            if (foo == 0 && (S.sig[s] & SIG_RED)) {
              // only reached with PIPAMD_T_NOSKIP: multipliers (1, 0) and gcd 1, the reference
              // rewrites the row with the bits it read -- so do we, without the arithmetic
              row_store<T, NCH>(r, row, ncolp, lane);
              if (lane == 0) S.sig[s] &= ~0xC0;
              continue;
            }
            // pivot > 0 (choisir_piv only takes positive entries): with pivot == 1, or
            // gcd(pivot, foo) == 1, the divisions of traiter.c:472-474 are by 1
            T den_s = uni64(S.den[s]);
            T d = 1, lp = pivot, g0 = den_s;
            const T foo_in = foo;
            for (int rep16 = 0; rep16 < PIP_DUP_REPS(16); rep16++) {
              if (PIP_DUP == 16) {
                PIP_OPAQUE_MEM();
                den_s = uni64(S.den[s]);
                foo = foo_in;
                if constexpr (sizeof(T) == 8) asm volatile("" : "+s"(foo));
                d = 1, lp = pivot, g0 = den_s;
              }
              if (pivot != 1) {
                d = gcd_i64(pivot, foo);
                if (d != 1) {
                  lp = exact_quo(pivot, d);
                  foo = exact_quo(foo, d);
                }
                g0 = wmul(lp, den_s);
              }
            }
            PROF(10);
            bool done_small = false;
            if constexpr (PIP_DUP == 17 && sizeof(T) == 8) {  // the row update once more, on a copy
              RowRegs<T, NCH> r2 = r;
              T nd2;
              asm volatile("" : "+v"(r2.v[0][0]));
              bool okk;
              const T lim = (T)1 << 15;
              if (S.rcls[s] == 0 && prow_cls == 0 && lp < lim && foo < lim && foo > -lim && dpiv < lim && dpiv > -lim)
                { RowRegs32<NCH> zz; okk = update_row_small<NCH>(r2, zz, S.prow, pivj, (int)lp, (int)foo, (int)dpiv, g0, lane, nd2); }
              else
                okk = update_row<T, NCH>(r2, S.prow, pivj, lp, foo, dpiv, g0, lane, nd2);
              asm volatile("" ::"v"(r2.v[0][0]), "v"(r2.v[0][1]), "s"(nd2), "s"((int)okk));
              PIP_OPAQUE_MEM();
            }
            if constexpr (sizeof(T) == 8) {
              // both rows in magnitude class 0 (entries below 2^15) and small multipliers: 32-bit path
              const T lim = (T)1 << 15;
              if (S.rcls[s] == 0 && prow_cls == 0 && lp < lim && foo < lim && foo > -lim && dpiv < lim && dpiv > -lim) {
                RowRegs32<NCH> z32;
                if (!update_row_small<NCH>(r, z32, S.prow, pivj, (int)lp, (int)foo, (int)dpiv, g0, lane, nd)) {
                  if (lane == 0) sc.bad = 1;
                }
                if (!has_parm) {
                  PROF(11);
                  row_store32<NCH>(z32, row, ncolp, lane);
                  row_publish32<NCH>(z32, S, s, nvar, pivj, SIG_RED, lane);
                  if (lane == 0) S.den[s] = nd;
                  PROF(12);
                  continue;
                }
#pragma unroll
                for (int c = 0; c < NCH; c++)
#pragma unroll
                  for (int h = 0; h < 2; h++) r.v[c][h] = (T)z32.v[c][h];
                done_small = true;
              }
            }
            if constexpr (sizeof(T) == 16) {
              // 128-bit entries whose operands are all below 2^31 (both rows in magnitude class 0) under a denominator
              // product of at most 63 bits: the update on 64-bit registers (update_row_narrow)
              const T lim = (T)1 << 31, glim = (T)1 << 62;
              if (PIP_OPT_NARROW128 && S.rcls[s] == 0 && prow_cls == 0 && lp < lim && foo < lim && foo > -lim && dpiv < lim &&
                  dpiv > -lim && g0 < glim && g0 > -glim) {
                if (!update_row_narrow<NCH>(r, S.prow, pivj, (i64)lp, (i64)foo, (i64)dpiv, (i64)g0, lane, nd)) {
                  if (lane == 0) sc.bad = 1;
                }
                done_small = true;
              }
            }
            if (!done_small && !update_row<T, NCH, (NW == 1 && NCH == 1 && sizeof(T) == 8)>(r, S.prow, pivj, lp, foo, dpiv, g0, lane, nd)) {
              if (lane == 0) sc.bad = 1;
            }
            PROF(11);
            PROF_CNT(prof, 0, true);
            PROF_CNT(prof, 1, pivot != 1);
            PROF_CNT(prof, 2, d != 1);
            PROF_CNT(prof, 3, g0 != 1);
            PROF_CNT(prof, 4, nd != g0);
            PROF_CNT(prof, 5, uni64(S.den[s]) != 1);
            PROF_CNT(prof, 6, (u64)uabs64(pivot) >> 16 != 0);
            for (int rep18 = 0; rep18 < PIP_DUP_REPS(18); rep18++) {
              if (PIP_DUP == 18) PIP_OPAQUE_MEM();
              row_store<T, NCH>(r, row, ncolp, lane);
              row_publish<T, NCH>(r, S, s, nvar, ncol, bigparm, pivj, SIG_RED, has_parm, lane);
            }
            if (lane == 0) S.den[s] = nd;
            PROF(12);
          }
        }
      }
    }
    bsync<NW>();
    PROF(6);
    if (sc.bad) {
      status = PIPAMD_ST_OVERFLOW;
      break;
    }
    // ---------------- C: swap roles, refresh the sign hints, next chercher ------
    if (tid == 0) {  // traiter.c:514-516: the pivot row becomes the unit row of column pivj
      S.ref[pivi] = (u16)(UNITBIT | UNITZERO | pivj);
      S.urow[pivj] = (u16)pivi;
    }
    for (int rep10 = 0; rep10 < PIP_DUP_REPS(10); rep10++)
    for (int s = tid; s < ni; s += NT) {
      if (PIP_DUP == 10) PIP_OPAQUE_MEM();
      int ff, k;
      if (s == pslot) {  // traiter.c:503-513: its slot now holds the row that replaces ku's unit row
        k = ku;
        ff = PIPAMD_F_PLUS;
        S.den[s] = pivot;
        S.srow[s] = (u16)ku;
        S.ref[ku] = (u16)s;
      } else {
        k = S.srow[s];
        ff = S.fl[s];
      }
      // traiter.c:518-529
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
      else if (ff == PIPAMD_F_UNKNOWN && bigparm < 0) {
        const int ec = exam_class(sg);
        S.nf[s] = (u8)ec;
        if (ec == PIPAMD_F_MINUS) atomicMin(&sc.pivi2, k);
      }
    }
    bsync<NW>();
    PROF(7);
  }

  // ---- epilogue: publish the row tables, the header and (if any) the solution
  bsync<NW>();
  for (int i = tid; i < nligne; i += NT) {
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
  tflags &= ~PIPAMD_T_STATE;
  if (status == PIPAMD_ST_RUN || status == PIPAMD_ST_NEED_COMPA) {
    // paused (pivot budget spent, LDS image full, or waiting for the host's sign tests, which
    // only touch flags): save the summaries for the launch that resumes the job
    for (int s = tid; s < ni; s += NT) {
      g_sig[s] = S.sig[s];
      g_rcls[s] = S.rcls[s];
    }
    for (int e = tid; e < ni * NM; e += NT) g_nzm[e] = S.nzm[e];
    tflags |= PIPAMD_T_STATE;
  }
  if (status == PIPAMD_ST_SOLUTION) {
    // solution(), traiter.c:255-271: rows 0..nvar-1, parameters then constant
    T *sol_num = (T *)(arena + J->sol_off);
    T *sol_den = sol_num + (size_t)nvar * (nparm + 1);
    for (int e = tid; e < nvar * (nparm + 1); e += NT) {
      int i = e / (nparm + 1), jj = e % (nparm + 1);
      int col = jj < nparm ? nvar + 1 + jj : nvar;
      const int rf = S.ref[i];
      T v = 0;
      if (!(rf & UNITBIT)) v = vals[(size_t)rf * W + col];
      sol_num[e] = v;
    }
    for (int i = tid; i < nvar; i += NT) {
      const int rf = S.ref[i];
      sol_den[i] = (rf & UNITBIT) ? 1 : S.den[rf];
    }
  }
  int mc = 0;
  if (wave == 0) {
    for (int s = lane; s < ni; s += 64)
      if (S.rcls[s] > mc) mc = S.rcls[s];
    mc = ballot64(mc == 3) ? 3 : (ballot64(mc == 2) ? 2 : (ballot64(mc == 1) ? 1 : 0));
  }
  if (tid == 0) {
    J->ni = ni;
    J->npiv = npiv;
    J->ncut = ncut;
    J->nupd = nupd;
    J->nlog = nlog;
    J->tflags = tflags;
    J->state_nch = NCH;
    J->maxabs = (u64)mc;  // magnitude class of the largest entry
    J->aux = sc.aux;
    J->status = status;
    if (status == PIPAMD_ST_RUN && q.out_count) {  // paused: the next launch resumes it
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, ni);
    }
    if (status == PIPAMD_ST_CAPACITY && q.out_count) {  // no spare row left: the host re-houses it (expanser)
      q.out_list[atomicAdd(q.out_count, 1)] = jb;
      atomicMax(q.out_maxni, PIPAMD_Q_CAPFLAG | ni);
      atomicAdd(q.out_maxni + 1, 1);  // (the list\'s third control word: tableaux out of rows)
    }
  }
  if constexpr (GM) {  // give the HBM image back
    __threadfence();
    __syncthreads();
    if (tid == 0) atomicExch((int *)gimg + gslot, 0);
  }
  PROF(8);
  PROF_FLUSH(prof);
}

// What a launch needs besides the jobs: its LDS image (Lmax, Smax, Wmax), the pivot budget per
// job, waves per job, entry width, the work queue and the number of workgroups.
struct AdvanceLaunch {
  PipJob *jobs;
  i64 *arena;
  int njobs, Lmax, Smax, Wmax, iter_limit;
  PipQueue q;
  int waves; // waves per job: 1, 4 or 8 (8: 64-bit entries of <= 128 columns only, else 4)
  int grid;  // workgroups = upper bound on the entries of the input list (0: njobs)
  bool full; // every job: no parameters, no big parameter, nvar + 1 == W == 128 (see FULL)
  unsigned long long *prof;
  size_t shm;
  unsigned char *gimg;  // HBM blocks for the row tables when they do not fit LDS (GM instantiation), else NULL
  int gslots;           // number of those blocks (behind as many lock words)
  hipStream_t stream;
};

// hipFuncSetAttribute applies to the current device only: remember per (instantiation, device)
// whether the opt-in to more than 48 KiB of dynamic LDS was made.
template <class T, int NCH, int NW, bool GM, int SC, bool FULL>
hipError_t launch_advance_t(const AdvanceLaunch &a) {
  const void *fn = (const void *)pip_advance_kernel<T, NCH, NW, GM, SC, FULL>;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!GM && a.shm > 48 * 1024) {  // large tableaux: opt in to more than the default dynamic LDS (160 KiB per CU)
    static std::atomic<unsigned long long> raised{0};
    if (!((raised.load(std::memory_order_acquire) >> dev) & 1)) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PIPAMD_LDS_BUDGET);
      if (e != hipSuccess) return e;
      raised.fetch_or(1ull << dev, std::memory_order_release);
    }
  }
  const int grid = a.grid > 0 && a.grid < a.njobs ? a.grid : a.njobs;
  hipLaunchKernelGGL((pip_advance_kernel<T, NCH, NW, GM, SC, FULL>), dim3(grid), dim3(64 * NW), GM ? 0 : a.shm, a.stream, a.jobs,
                     a.arena, a.njobs, a.Lmax, a.Smax, a.Wmax, a.iter_limit, a.q, a.gimg, a.shm, a.gslots, a.prof);
  return hipGetLastError();
}

#include "pip_adv_inst.h"
#endif  // PIP_ADVANCE_H
