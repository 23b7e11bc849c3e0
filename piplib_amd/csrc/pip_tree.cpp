// piplib_amd/csrc/pip_tree.cpp -- layer 3 of the C ABI: PipLib front-end semantics with the
// quast decision tree on the host and every pivot on the GPU.
//
// The host owns exactly what the reference's traiter() does *between* pivot runs
// (source/traiter.c:628-791):
//   * compa_test (traiter.c:162-243): the sign tests of undecided rows are integer
//     feasibility problems over the context; all of them are built here and solved as one
//     batch of device jobs, then applied in the reference's order (stop at the first row
//     proven negative);
//   * the tree split on a Critic/Unknown row (traiter.c:695-759): device-to-device copy of
//     the tableau for the "then" branch, context rows kept on the host;
//   * parametric Gomory cuts (integrer.c:487-520, find_parm/add_parm integrer.c:156-291):
//     the context and the solution tape live here, the cut row is appended to the device
//     tableau;
//   * the solution tape (sol.c:52-226) and its sol_edit text (sol.c:291-422).
// Everything that pivots -- including the context-emptiness test of the front ends
// (maind.c:196-203) and every compa_test sub-problem -- runs in pip_advance_kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "pip_host.h"
#include "pip_quast.h"

namespace {

typedef long long i64;
typedef unsigned long long u64;

#define HIPTHROW(call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      pipamd_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      throw (int)PIPAMD_E_HIP;                                                                      \
    }                                                                                               \
  } while (0)

// ---- wrap-around integer helpers (piplib.h:128-169, integrer.c:43-74), host side, for both
// entry widths: E = long long (the reference's int64 build) or __int128 (the overflow-safe one)
typedef __int128 i128;
template <class E> struct UT;
template <> struct UT<i64> { typedef u64 type; };
template <> struct UT<i128> { typedef unsigned __int128 type; };
template <class E> inline E wadd(E a, E b) { return (E)((typename UT<E>::type)a + (typename UT<E>::type)b); }
template <class E> inline E wsub(E a, E b) { return (E)((typename UT<E>::type)a - (typename UT<E>::type)b); }
template <class E> inline E wneg(E a) { return (E)((typename UT<E>::type)0 - (typename UT<E>::type)a); }
template <class E> inline E wabs(E a) { return a < 0 ? wneg(a) : a; }
template <class E> inline E crem(E a, E b) { return (b == 0 || b == -1) ? (E)0 : a % b; }
template <class E> inline E cquo(E a, E b) { return b == 0 ? (E)0 : (b == -1 ? wneg(a) : a / b); }
template <class E> E gcd(E a, E b) {
  while (b) {
    E t = crem(a, b);
    a = b;
    b = t;
  }
  return wabs(a);
}
template <class E> inline E fmod_(E a, E b) {
  E m = crem(a, b);
  if (m < 0) m = wadd(m, wabs(b));
  return m;
}
template <class E> inline E floordiv(E a, E b) { return cquo(wsub(a, fmod_(a, b)), b); }
template <class E> inline E wmul(E a, E b) { return (E)((typename UT<E>::type)a * (typename UT<E>::type)b); }

enum { S_FREE = 0, S_NIL, S_IF, S_LIST, S_FORM, S_NEW, S_DIV, S_VAL }; /* sol.c:42-50 */
template <class E> struct CellT {
  int kind;
  E a, b;
};
typedef CellT<i64> Cell;

// context: rows of (parameters | constant), traiter.c keeps it as a Tableau
template <class E> struct CtxT {
  int nc = 0, width = 0;  // rows in use, columns allocated per row
  std::vector<E> v;
  E &at(int r, int c) { return v[(size_t)r * width + c]; }
  E at(int r, int c) const { return v[(size_t)r * width + c]; }
  void reserve(int rows, int cols) {
    if (cols > width) {
      std::vector<E> nv((size_t)std::max(rows, nc + 4) * cols, 0);
      for (int r = 0; r < nc; r++)
        for (int c = 0; c < width; c++) nv[(size_t)r * cols + c] = v[(size_t)r * width + c];
      v.swap(nv);
      width = cols;
    }
    if ((size_t)rows * width > v.size()) v.resize((size_t)(rows + 8) * width, 0);
  }

  // ---- parameters introduced by parametric cuts (integrer.c:156-291) ----
  // A cut with parameter part c.p + c0 over the denominator D needs q = floor(-(c.p + c0) / D),
  // i.e. the parameter q with 0 <= -(c.p + c0) - D q <= D - 1.  The context stores such a
  // definition as the two rows of that double inequality, with q in a column of its own:
  //     lower:  -c.p - D q - c0          >= 0
  //     upper:   c.p + D q + c0 + D - 1  >= 0
  struct Quotient {
    std::vector<E> c;  // coefficients of the nparm existing parameters
    E c0, D;           // constant, divisor
  };
  // is there a row  sign * (c on the columns before p | D in column p | 0 behind it)  with constant cst?
  bool has_row(int nparm, int p, const Quotient &k, bool negated, E cst) const {
    for (int r = 0; r < nc; r++) {
      if (at(r, p) != (negated ? wneg(k.D) : k.D) || at(r, nparm) != cst) continue;
      bool same = true;
      for (int col = p + 1; col < nparm && same; col++) same = at(r, col) == 0;
      for (int col = 0; col < p && same; col++) same = at(r, col) == (negated ? wneg(k.c[col]) : k.c[col]);
      if (same) return true;
    }
    return false;
  }
  // A parameter the cut does not use (trailing zero coefficients) whose two defining rows are the
  // ones this quotient would get: its column, or -1 (find_parm, integrer.c:258-291).
  int find_quotient(int nparm, const Quotient &k) const {
    const E upper_cst = wsub(wadd(k.c0, k.D), (E)1);
    for (int p = nparm - 1; p >= 0 && k.c[p] == 0; --p)
      if (has_row(nparm, p, k, false, upper_cst) && has_row(nparm, p, k, true, wneg(k.c0))) return p;
    return -1;
  }
  // The new parameter takes column nparm (the constants move one column right) and its two rows
  // are appended (add_parm, integrer.c:194-224).  Returns its column.
  int define_quotient(int nparm, const Quotient &k) {
    const int nr = nc;
    reserve(nr + 2, nparm + 2);
    for (int r = 0; r < nr; r++) {
      at(r, nparm + 1) = at(r, nparm);
      at(r, nparm) = 0;
    }
    for (int j = 0; j < nparm; j++) {
      at(nr, j) = wneg(k.c[j]);
      at(nr + 1, j) = k.c[j];
    }
    at(nr, nparm) = wneg(k.D);
    at(nr + 1, nparm) = k.D;
    at(nr, nparm + 1) = wneg(k.c0);
    at(nr + 1, nparm + 1) = wadd(wsub(k.c0, (E)1), k.D);
    nc += 2;
    return nparm;
  }
  // what the solution tape says about it (integrer.c:173-188): New k, Div, the form -c.p - c0, D
  template <class PUSH>
  static void announce_quotient(PUSH &&push, int nparm, const Quotient &k) {
    push(S_NEW, (E)nparm, (E)0);
    push(S_DIV, (E)0, (E)0);
    push(S_FORM, (E)(nparm + 1), (E)0);
    for (int j = 0; j < nparm; j++) push(S_VAL, wneg(k.c[j]), (E)1);
    push(S_VAL, wneg(k.c0), (E)1);
    push(S_VAL, k.D, (E)1);
  }
};
typedef CtxT<i64> Ctx;

struct HostJob {
  PipJob pj;
  i64 block_off = 0;
  size_t block_words = 0;
};

template <class E> struct SnapT {  // host copy of a job's row tables and rows
  int L = 0, S = 0, W = 0;
  std::vector<E> den, vals;
  std::vector<int> flag, ref;
  const E *row(int k) const { return &vals[(size_t)ref[k] * W]; }
};

template <class E>
class TreeT {
 public:
  typedef CtxT<E> Ctx;
  typedef CellT<E> Cell;
  typedef SnapT<E> Snap;
  static constexpr int EW = (int)(sizeof(E) / 8);   // int64 words per entry
  static constexpr int EBITS = (int)(sizeof(E) * 8);
  TreeT(pipamd_engine *e, int deepest) : deepest_(deepest) {
    (void)e;
    const hipError_t err = hipStreamCreateWithFlags(&st_, hipStreamNonBlocking);
    if (err != hipSuccess) {  // never fall back to the null stream: it would serialise every tree of the process
      st_ = 0;
      pipamd_set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(err));
      throw (int)PIPAMD_E_HIP;
    }
  }
  ~TreeT() {
    if (d_arena_) hipFree(d_arena_);
    if (d_jobs_) hipFree(d_jobs_);
    if (d_big_) hipFree(d_big_);
    if (st_) hipStreamDestroy(st_);
  }
  // reuse the device buffers for another problem
  void reset(int deepest) {
    deepest_ = deepest;
    dual_ = false;
    tape.clear();
    pivots = 0;
    fail_status = 0;
    top_ = 0;
  }
  std::vector<Cell> tape;
  bool dual_ = false;  // Compute_dual (rational solves only, piplib.c:854-857)
  long long pivots = 0;
  int fail_status = 0;

  // traiter() itself (traiter.c:628): one call on a freshly built tableau and context, what
  // piplib.c:858 and maind.c:205 hand to it; the tape receives what traiter would have emitted
  void traiter_call(int nvar, int nparm, int ni, int nc, int bigparm, int tfl, const i64 *ineq, const i64 *ctxrows) {
    Ctx ctx;
    ctx.reserve(nc + 4, nparm + 2);
    ctx.nc = nc;
    for (int r = 0; r < nc; r++)
      for (int c = 0; c <= nparm; c++) ctx.at(r, c) = (E)ctxrows[(size_t)r * (nparm + 1) + c];
    dual_ = (tfl & PIPAMD_T_DUAL) != 0;
    HostJob job = make_job(nvar, nparm, ni, bigparm, tfl, ineq);
    node(job, ctx, nvar, nparm, ni, bigparm, tfl);
  }

  // maind.c:196-231: context emptiness test, then traiter
  bool front(int nvar, int nparm, int ni, int nc, int bigparm, int nq, const i64 *ineq, const i64 *ctxrows) {
    Ctx ctx;
    ctx.reserve(nc + 4, nparm + 2);
    ctx.nc = nc;
    for (int r = 0; r < nc; r++)
      for (int c = 0; c <= nparm; c++) ctx.at(r, c) = (E)ctxrows[(size_t)r * (nparm + 1) + c];
    if (nc) {
      size_t mark = top_;
      HostJob cj = make_context_job(ctx, nparm, nc, nullptr);
      std::vector<HostJob *> js{&cj};
      run_to_final(js);
      pivots += cj.pj.npiv;
      check_final(cj);
      top_ = mark;
      if (cj.pj.status == PIPAMD_ST_NIL) return false;  // "void"
    }
    const int tfl = nq ? PIPAMD_T_INT : (dual_ ? PIPAMD_T_DUAL : 0);
    HostJob job = make_job(nvar, nparm, ni, bigparm, tfl, ineq);
    node(job, ctx, nvar, nparm, ni, bigparm, tfl);
    return true;
  }

 public:  // (internal class; the lock-step Forest below reuses its helpers)
  int deepest_;
  bool speculative_ = false;           // run(): bounded effort (compa_test sub-problems that may never be needed)
  enum { SPEC_PIVOTS = 256 };
  hipStream_t st_ = 0;
  i64 *d_arena_ = nullptr;
  size_t arena_words_ = 0, top_ = 0;
  PipJob *d_jobs_ = nullptr;
  // row tables of jobs that outgrow LDS (the launcher sizes it): {buffer, bytes}
  void *d_big_ = nullptr;
  size_t big_bytes_ = 0;
  void *big_[2] = {&d_big_, &big_bytes_};
  int d_jobs_cap_ = 0;

  void fail(int st) {
    fail_status = st;
    throw (int)PIPAMD_E_SOLVER;
  }
  // every transfer and launch of a tree goes through its own stream, so several trees (one per
  // host thread) overlap on the GPU
  void copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    HIPTHROW(hipMemcpyAsync(dst, src, bytes, kind, st_));
    HIPTHROW(hipStreamSynchronize(st_));
  }
  void push(int kind, E a, E b) {
    tape.push_back(Cell{kind, a, b});
    if (tape.size() >= 4096) fail(PIPAMD_ST_INTERNAL);  // "The solution is too complex", sol.c:97
  }

  // ---------------------------------------------------------------- arena
  void ensure_arena(size_t words) {
    if (words <= arena_words_) return;
    size_t nw = std::max(words * 2, (size_t)1 << 20);
    i64 *n = nullptr;
    HIPTHROW(hipMalloc((void **)&n, nw * sizeof(i64)));
    if (d_arena_) {
      copy(n, d_arena_, top_ * sizeof(i64), hipMemcpyDeviceToDevice);
      hipFree(d_arena_);
    }
    d_arena_ = n;
    arena_words_ = nw;
  }
  static int even(int x) { return (x + 1) & ~1; }
  static bool rows_fit(int nvar, int S, int W) {
    if (S > PIPAMD_SMAX || even(nvar + S) > PIPAMD_LMAX) return false;
    // 64-bit: row tables that outgrow LDS live in HBM, only the 16-bit row codes bound a job;
    // 128-bit: the tables must fit a workgroup's LDS
    return EW == 1 || pipk_advance_lds_bytes((even(nvar + S) + 3) & ~3, (S + 3) & ~3, W, EBITS) <= PIPAMD_LDS_BUDGET;
  }
  // Row capacity of the block a job that ran out of rows is re-housed in: geometric growth (a
  // sub-problem may need thousands of cut rows, and every re-housing copies the whole tableau),
  // clamped to the engine's row limit -- S+1 when not even that fits, which alloc_job then
  // refuses with PIPAMD_ST_CAPACITY.
  static int next_rows(int nvar, int S, int W) {
    const int want = std::max(S + 32, S * 3 / 2);
    if (rows_fit(nvar, want, W) || !rows_fit(nvar, S + 1, W)) return rows_fit(nvar, want, W) ? want : S + 1;
    int lo = S + 1, hi = want;  // lo fits, hi does not
    while (hi - lo > 1) {
      const int mid = lo + (hi - lo) / 2;
      (rows_fit(nvar, mid, W) ? lo : hi) = mid;
    }
    return lo;
  }

  HostJob alloc_job(int nvar, int nparm, int ni, int bigparm, int tflags, int S, int W) {
    HostJob j;
    memset(&j.pj, 0, sizeof j.pj);
    S = std::max(S, ni + 1);
    W = even(std::max(W, nvar + nparm + 1));
    if (W > PIPAMD_MAXCOL || !rows_fit(nvar, S, W)) {
      if (getenv("PIPAMD_TREE_TRACE"))
        fprintf(stderr, "[tree] job beyond the engine's limits: nvar %d nparm %d ni %d S %d W %d\n", nvar, nparm, ni, S, W);
      fail(PIPAMD_ST_CAPACITY);
    }
    const int L = even(nvar + S);
    const int nm = 8;  // room for the bitmaps of any launch geometry (jobs of mixed widths share launches)
    const size_t sol = (size_t)even(nvar * (W - nvar) + nvar) * EW;
    const size_t state = (size_t)even(S * nm + (3 * L + 7) / 8) + 2 * PIPAMD_DETLOG * EW;  // summaries | determinant log
    // den[L] (entries) | flag[L] | ref[L], then S x W entries, the solution, the saved summaries
    j.block_words = (size_t)(EW + 1) * L + (size_t)S * W * EW + sol + state;
    ensure_arena(top_ + j.block_words);
    j.block_off = (i64)top_;
    top_ += j.block_words;
    j.pj.rows_off = j.block_off;
    j.pj.vals_off = j.block_off + (i64)(EW + 1) * L;
    j.pj.sol_off = j.pj.vals_off + (i64)S * W * EW;
    j.pj.state_off = j.pj.sol_off + (i64)sol;
    j.pj.log_off = j.pj.state_off + (i64)state - 2 * PIPAMD_DETLOG * EW;
    j.pj.nvar = nvar;
    j.pj.nparm = nparm;
    j.pj.ni = ni;
    j.pj.bigparm = bigparm;
    j.pj.tflags = tflags | PIPAMD_T_SORT | (deepest_ ? PIPAMD_T_DEEPEST : 0);
    j.pj.L = L;
    j.pj.S = S;
    j.pj.W = W;
    j.pj.status = PIPAMD_ST_RUN;
    j.pj.ldet = 1;
    j.pj.det[0] = 1;
    j.pj.ebits = EBITS;
    return j;
  }

  // bytes of a job's row tables + rows, and typed views into a host copy of them
  static size_t tab_words(int L, int S, int W) { return (size_t)(EW + 1) * L + (size_t)S * W * EW; }
  static E *blk_den(std::vector<i64> &blk) { return (E *)blk.data(); }
  static int *blk_flag(std::vector<i64> &blk, int L) { return (int *)(blk_den(blk) + L); }
  static E *blk_vals(std::vector<i64> &blk, int L) { return (E *)(blk.data() + (size_t)(EW + 1) * L); }

  // tab_alloc + tab_get (tab.c:158-248): nvar unit rows, ni Unknown rows with denominator 1
  void upload_fresh(HostJob &j, const std::vector<E> &rows /* ni x ncol */) {
    const int nvar = j.pj.nvar, ni = j.pj.ni, ncol = nvar + j.pj.nparm + 1, L = j.pj.L, S = j.pj.S, W = j.pj.W;
    std::vector<i64> blk(tab_words(L, S, W), 0);
    E *den = blk_den(blk), *vals = blk_vals(blk, L);
    int *flag = blk_flag(blk, L), *ref = flag + L;
    for (int i = 0; i < nvar; i++) {
      den[i] = 1;
      flag[i] = PIPAMD_F_UNIT;
      ref[i] = i;
    }
    for (int i = 0; i < ni; i++) {
      den[nvar + i] = 1;
      flag[nvar + i] = PIPAMD_F_UNKNOWN;
      ref[nvar + i] = i;
      for (int c = 0; c < ncol; c++) vals[(size_t)i * W + c] = rows[(size_t)i * ncol + c];
    }
    copy(d_arena_ + j.block_off, blk.data(), blk.size() * sizeof(i64), hipMemcpyHostToDevice);
  }

  HostJob make_job(int nvar, int nparm, int ni, int bigparm, int tflags, const i64 *rows) {
    const int ncol = nvar + nparm + 1;
    HostJob j = alloc_job(nvar, nparm, ni, bigparm, tflags, ni + 24, ncol + (nparm ? 6 : 0));
    std::vector<E> r((size_t)ni * ncol);
    for (size_t k = 0; k < r.size(); k++) r[k] = (E)rows[k];
    upload_fresh(j, r);
    return j;
  }
  // expanser(context, nparm, nc, nparm+1, nparm, extra?1:0, 0) (traiter.c:191,211,
  // maind.c:198): the context as a problem in the parameters, optionally one more row
  HostJob make_context_job(const Ctx &ctx, int nparm, int nc, const std::vector<E> *extra) {
    const int ni = nc + (extra ? 1 : 0), ncol = nparm + 1;
    HostJob j = alloc_job(nparm, 0, ni, -1, PIPAMD_T_INT, ni + 16, ncol);
    std::vector<E> r((size_t)ni * ncol);
    for (int i = 0; i < nc; i++)
      for (int c = 0; c < ncol; c++) r[(size_t)i * ncol + c] = ctx.at(i, c);
    if (extra)
      for (int c = 0; c < ncol; c++) r[(size_t)nc * ncol + c] = (*extra)[c];
    upload_fresh(j, r);
    return j;
  }

  Snap download(const HostJob &j) {
    Snap s;
    s.L = j.pj.L;
    s.S = j.pj.S;
    s.W = j.pj.W;
    std::vector<i64> blk(tab_words(s.L, s.S, s.W));
    copy(blk.data(), d_arena_ + j.block_off, blk.size() * sizeof(i64), hipMemcpyDeviceToHost);
    const E *den = blk_den(blk), *vals = blk_vals(blk, s.L);
    s.den.assign(den, den + s.L);
    const int *flag = blk_flag(blk, s.L);
    s.flag.assign(flag, flag + s.L);
    s.ref.assign(flag + s.L, flag + 2 * s.L);
    s.vals.assign(vals, vals + (size_t)s.S * s.W);
    return s;
  }
  void set_flag(const HostJob &j, int row, int f) {
    int *g_flag = (int *)((E *)(d_arena_ + j.pj.rows_off) + j.pj.L);
    copy(g_flag + row, &f, sizeof(int), hipMemcpyHostToDevice);
  }

  // run the engine on a set of jobs until none is PIPAMD_ST_RUN
  void run(std::vector<HostJob *> &js) {
    const int n = (int)js.size();
    if (!n) return;
    if (n > d_jobs_cap_) {
      if (d_jobs_) hipFree(d_jobs_);
      d_jobs_cap_ = n + 64;
      HIPTHROW(hipMalloc((void **)&d_jobs_, sizeof(PipJob) * d_jobs_cap_));
    }
    std::vector<PipJob> tab(n);
    int Lm = 4, Sm = 4, Wm = 2;
    for (int i = 0; i < n; i++) {
      tab[i] = js[i]->pj;
      Lm = std::max(Lm, (int)tab[i].L);
      Sm = std::max(Sm, (int)tab[i].S);
      Wm = std::max(Wm, (int)tab[i].W);
    }
    if (speculative_) {
      // bounded effort: one launch of at most SPEC_PIVOTS pivots per job, no re-housing; jobs
      // that are not done stay PIPAMD_ST_RUN / PIPAMD_ST_CAPACITY for a later, unbounded run
      copy(d_jobs_, tab.data(), sizeof(PipJob) * n, hipMemcpyHostToDevice);
      HIPTHROW(pipk_launch_advance_q(d_jobs_, d_arena_, n, Lm, Sm, Wm, SPEC_PIVOTS, n >= 2048 ? 1 : 4, EBITS, nullptr, 0, big_, 0,
                                     nullptr, st_));
      copy(tab.data(), d_jobs_, sizeof(PipJob) * n, hipMemcpyDeviceToHost);
      for (int i = 0; i < n; i++) js[i]->pj = tab[i];
      return;
    }
    for (int pass = 0; pass < 64; pass++) {
      copy(d_jobs_, tab.data(), sizeof(PipJob) * n, hipMemcpyHostToDevice);
      for (int guard = 0; guard < 4096; guard++) {
        HIPTHROW(pipk_launch_advance_q(d_jobs_, d_arena_, n, Lm, Sm, Wm, 1 << 20, n >= 2048 ? 1 : 4, EBITS, nullptr, 0, big_, 0,
                                       nullptr, st_));
        copy(tab.data(), d_jobs_, sizeof(PipJob) * n, hipMemcpyDeviceToHost);
        bool again = false;
        for (int i = 0; i < n; i++)
          if (tab[i].status == PIPAMD_ST_RUN) again = true;
        if (!again) break;
      }
      // a tableau that ran out of spare rows is re-housed in a larger block (expanser) and resumed
      bool grown = false;
      for (int i = 0; i < n; i++) {
        js[i]->pj = tab[i];
        if (tab[i].status == PIPAMD_ST_CAPACITY) {
          grow(*js[i], next_rows(js[i]->pj.nvar, js[i]->pj.S, js[i]->pj.W), js[i]->pj.W);
          tab[i] = js[i]->pj;
          Lm = std::max(Lm, (int)tab[i].L);
          Sm = std::max(Sm, (int)tab[i].S);
          grown = true;
        }
      }
      if (!grown) break;
    }
  }
  // non-parametric jobs (context test, compa_test sub-problems): run to a final status.  With
  // the deepest-cut option the kernel hands every cut to the host (integrer.c:417-438).
  void run_to_final(std::vector<HostJob *> &js) {
    for (int guard = 0; guard < 100000; guard++) {
      run(js);
      if (speculative_) return;
      bool again = false;
      for (HostJob *j : js)
        if (j->pj.status == PIPAMD_ST_NEED_PARMCUT) {
          Ctx none;
          int np = 0, ni = j->pj.ni;
          if (host_cut(*j, none, j->pj.nvar, np, ni, -1)) {
            j->pj.status = PIPAMD_ST_RUN;
            again = true;
          } else
            j->pj.status = PIPAMD_ST_NIL;
        }
      if (!again) return;
    }
  }
  // statuses that end a traiter() call abnormally (the reference exits the process)
  void check_final(const HostJob &j) {
    const int s = j.pj.status;
    if (s == PIPAMD_ST_OVERFLOW || s == PIPAMD_ST_RANGE || s == PIPAMD_ST_INTERNAL || s == PIPAMD_ST_MAXCOL ||
        s == PIPAMD_ST_RUN)
      fail(s);
  }

  // Re-house a job in a larger block (more rows / columns): expanser (traiter.c:55-88)
  void grow(HostJob &j, int newS, int newW) {
    Snap s = download(j);
    HostJob n = alloc_job(j.pj.nvar, j.pj.nparm, j.pj.ni, j.pj.bigparm, 0, newS, newW);
    const int L = n.pj.L, W = n.pj.W, nl = j.pj.nvar + j.pj.ni;
    std::vector<i64> blk(tab_words(L, n.pj.S, W), 0);
    E *den = blk_den(blk), *vals = blk_vals(blk, L);
    int *flag = blk_flag(blk, L), *ref = flag + L;
    for (int k = 0; k < nl; k++) {
      den[k] = s.den[k];
      flag[k] = s.flag[k];
      ref[k] = s.ref[k];
      if (!(s.flag[k] & PIPAMD_F_UNIT))
        for (int c = 0; c < s.W; c++) vals[(size_t)s.ref[k] * W + c] = s.vals[(size_t)s.ref[k] * s.W + c];
    }
    copy(d_arena_ + n.block_off, blk.data(), blk.size() * sizeof(i64), hipMemcpyHostToDevice);
    PipJob keep = j.pj;
    j.block_off = n.block_off;
    j.block_words = n.block_words;
    j.pj = n.pj;
    j.pj.tflags = keep.tflags & ~PIPAMD_T_STATE;  // summaries are rebuilt by the next launch
    j.pj.npiv = keep.npiv;
    j.pj.ncut = keep.ncut;
    j.pj.nupd = keep.nupd;
    j.pj.ldet = keep.ldet;
    memcpy(j.pj.det, keep.det, sizeof keep.det);
    j.pj.status = PIPAMD_ST_RUN;
  }

  // ------------------------------------------------------------ compa_test
  void compa(HostJob &job, const Ctx &ctx, int nvar, int nparm, int ni) {
    if (nparm == 0) return;
    if (nparm >= PIPAMD_MAXPARM) fail(PIPAMD_ST_INTERNAL);  // "Too much parameters"
    Snap s = download(job);
    const int nc = ctx.nc;
    std::vector<int> rows;
    for (int i = 0; i < ni + nvar; i++)
      if (s.flag[i] & (PIPAMD_F_CRITIC | PIPAMD_F_UNKNOWN)) rows.push_back(i);
    if (rows.empty()) return;
    const size_t mark = top_;
    std::vector<HostJob> sub;
    std::vector<int> critic(rows.size());
    sub.reserve(2 * rows.size());
    for (size_t t = 0; t < rows.size(); t++) {
      const E *r = s.row(rows[t]);
      int cr = 1;
      for (int j = 0; j < nvar; j++)
        if (r[j] > 0) {
          cr = 0;
          break;
        }
      critic[t] = cr;
      std::vector<E> ex(nparm + 1);
      for (int j = 0; j < nparm; j++) ex[j] = r[j + nvar + 1];  // "row >= 1" (>= 0 if critical)
      ex[nparm] = cr ? r[nvar] : wsub(r[nvar], (E)1);
      sub.push_back(make_context_job(ctx, nparm, nc, &ex));
      for (int j = 0; j < nparm; j++) ex[j] = wneg(r[j + nvar + 1]);  // "-row >= 1"
      ex[nparm] = wsub(wneg(r[nvar]), (E)1);
      sub.push_back(make_context_job(ctx, nparm, nc, &ex));
    }
    std::vector<HostJob *> ptr;
    for (auto &h : sub) ptr.push_back(&h);
    // All sub-problems at once, but with bounded effort: the reference only solves the rows up to
    // the first negative one, and a row behind it may be arbitrarily hard (or make the cut
    // generator run forever) without the reference ever noticing.
    speculative_ = true;
    run_to_final(ptr);
    speculative_ = false;
    // apply in the reference's order; sub-problems behind the first negative row were
    // speculative and are ignored (their pivots are not counted either).  A sub-problem the
    // bounded run did not finish is finished now that it is known to be needed.
    for (size_t t = 0; t < rows.size(); t++) {
      HostJob &jp = sub[2 * t], &jm = sub[2 * t + 1];
      std::vector<HostJob *> need;
      for (HostJob *h : {&jp, &jm})
        if (h->pj.status == PIPAMD_ST_RUN || h->pj.status == PIPAMD_ST_CAPACITY || h->pj.status == PIPAMD_ST_NEED_PARMCUT)
          need.push_back(h);
      if (!need.empty()) run_to_final(need);
      pivots += jp.pj.npiv + jm.pj.npiv;
      check_final(jp);
      check_final(jm);
      const bool cp = jp.pj.status != PIPAMD_ST_NIL, cm = jm.pj.status != PIPAMD_ST_NIL;
      int f;
      if (cp && cm)
        f = critic[t] ? PIPAMD_F_CRITIC : PIPAMD_F_UNKNOWN;
      else if (cm)
        f = PIPAMD_F_MINUS;
      else
        f = cp ? PIPAMD_F_PLUS : PIPAMD_F_ZERO;
      set_flag(job, rows[t], f);
      if (f == PIPAMD_F_MINUS) break;
    }
    top_ = mark;
  }

  // (find_parm / add_parm, integrer.c:156-291: CtxT::find_quotient / define_quotient)
  static E bezout(E x, E y, E delta) {  // integrer.c:98-150
    E a = 1, b = 0, c = 0, d = 1, u = y, v = delta;
    for (;;) {
      E q = floordiv(u, v), r = fmod_(u, v);
      if (r == 0) break;
      u = v;
      v = r;
      E e = wsub(a, wmul(q, c)), f = wsub(b, wmul(q, d));
      a = c;
      b = d;
      c = e;
      d = f;
    }
    if (v != 1) return 0;
    return fmod_(wmul(c, x), delta);
  }

  // integrer() for the row the kernel stopped at (integrer.c:342-520): the kernel hands over
  // parametric cuts, and every cut when the deepest-cut option is on.
  // Returns false when the row admits no cut (case (b): no solution).
  bool host_cut(HostJob &job, Ctx &ctx, int nvar, int &nparm, int &ni, int bigparm) {
    const int ci = job.pj.aux;
    Snap s = download(job);
    const int ncol = nvar + nparm + 1, nligne = nvar + ni;
    if (ncol >= PIPAMD_MAXCOL) fail(PIPAMD_ST_MAXCOL);
    const E *r = s.row(ci);
    const E D = s.den[ci];
    std::vector<E> cut(ncol + 1);
    bool ok_var = false, ok_parm = false;
    for (int j = 0; j < nvar; j++) {
      cut[j] = fmod_(r[j], D);
      if (cut[j] > 0) ok_var = true;
    }
    cut[nvar] = wneg(fmod_(wneg(r[nvar]), D));
    for (int j = nvar + 1; j < ncol; j++) {
      if (j == bigparm) {
        cut[j] = 0;
        continue;
      }
      cut[j] = wneg(fmod_(wneg(r[j]), D));
      if (cut[j] != 0) ok_parm = true;
    }
    cut[ncol] = D;
    int newcol = -1;
    if (!ok_parm) {
      if (!ok_var) return false;
      if (deepest_) {  // integrer.c:417-438
        E t = wneg(cut[nvar]), delta = gcd(t, D), tau = cquo(t, delta), dd = cquo(D, delta);
        t = wsub(dd, (E)1);
        E lambda = bezout(t, tau, dd);
        t = gcd(lambda, D);
        while (t != 1) {
          lambda = wadd(lambda, dd);
          t = gcd(lambda, D);
        }
        for (int j = 0; j < nvar; j++) cut[j] = fmod_(wmul(lambda, cut[j]), D);
        t = fmod_(wmul(cut[nvar], lambda), D);
        t = wsub(D, t);
        cut[nvar] = wneg(t);
      }
    } else {
      typename Ctx::Quotient k{std::vector<E>(cut.begin() + nvar + 1, cut.begin() + ncol), cut[nvar], D};
      int parm = ctx.find_quotient(nparm, k);
      if (parm == -1) {
        Ctx::announce_quotient([&](int kind, E a, E b) { push(kind, a, b); }, nparm, k);
        parm = ctx.define_quotient(nparm, k);
        nparm++;
      }
      if (!ok_var) fail(PIPAMD_ST_INTERNAL);  // assert(ok_var), integrer.c:499
      newcol = nvar + 1 + parm;
    }
    // append the cut as logical row nligne in slot ni
    const int need_w = nvar + nparm + 1;
    if (ni >= job.pj.S || nligne >= job.pj.L || need_w > job.pj.W)
      grow(job, std::max((int)job.pj.S, ni + 1) + 16, std::max((int)job.pj.W, need_w) + 4);
    const int W = job.pj.W, L = job.pj.L;
    std::vector<E> row(W, 0);
    for (int j = 0; j < ncol; j++) row[j] = cut[j];
    if (newcol >= 0) row[newcol] = wadd(row[newcol], cut[ncol]);
    copy((E *)(d_arena_ + job.pj.vals_off) + (size_t)ni * W, row.data(), W * sizeof(E), hipMemcpyHostToDevice);
    E *g_den = (E *)(d_arena_ + job.pj.rows_off);
    int *g_flag = (int *)(g_den + L), *g_ref = g_flag + L;
    const int fl = PIPAMD_F_MINUS;
    copy(g_den + nligne, &D, sizeof(E), hipMemcpyHostToDevice);
    copy(g_flag + nligne, &fl, sizeof(int), hipMemcpyHostToDevice);
    copy(g_ref + nligne, &ni, sizeof(int), hipMemcpyHostToDevice);
    ni++;
    job.pj.ni = ni;
    job.pj.nparm = nparm;
    job.pj.ncut++;
    job.pj.tflags &= ~PIPAMD_T_STATE;
    return true;
  }

  void emit_solution(const HostJob &job, int nvar, int nparm) {  // traiter.c:255-271
    const size_t n = (size_t)nvar * (nparm + 1);
    std::vector<E> buf(n + nvar);
    if (n + nvar)
      copy(buf.data(), d_arena_ + job.pj.sol_off, (n + nvar) * sizeof(E), hipMemcpyDeviceToHost);
    push(S_LIST, (E)nvar, 0);
    for (int i = 0; i < nvar; i++) {
      push(S_FORM, (E)(nparm + 1), 0);
      for (int j = 0; j <= nparm; j++) push(S_VAL, buf[(size_t)i * (nparm + 1) + j], buf[n + i]);
    }
  }

  // tab_sort_rows' bookkeeping for the dual (traiter.c:556-623): where each inequality of this
  // call's tableau ends up after the sort the kernel is about to do.  Replayed on the host from
  // a snapshot (same keys, same selection sort); rows do not move afterwards.
  static int trunc_x86(double t) { return (!(t > -2147483649.0 && t < 2147483648.0)) ? (int)0x80000000 : (int)t; }
  std::vector<int> dual_positions(const HostJob &job, int nvar, int ni) {
    Snap s = download(job);
    const int nligne = nvar + ni;
    std::vector<float> size(nligne, 0.f);
    std::vector<int> ineq(nligne, 0), order(nligne), pos(std::max(1, ni), 0);
    for (int i = 0; i < nligne; i++) order[i] = i;
    double smax = 0;
    for (int i = nvar; i < nligne; i++) {
      if (s.flag[i] & PIPAMD_F_UNIT) continue;
      const E *r = s.row(i);
      double sz = 0, d = (double)s.den[i];
      for (int j = 0; j < nvar; j++) {
        int q = trunc_x86((double)r[j] / d);
        double a = (double)(q < 0 ? (int)(0u - (unsigned)q) : q);
        sz = sz > a ? sz : a;
      }
      size[i] = (float)sz;
      smax = sz > smax ? sz : smax;
      ineq[i] = i - nvar;
    }
    std::vector<int> isunit(nligne);
    for (int i = 0; i < nligne; i++) isunit[i] = (s.flag[i] & PIPAMD_F_UNIT) != 0;
    for (int i = nvar; i < nligne; i++) {
      if (isunit[i]) continue;
      double sm = smax;
      int pv = i;
      for (int j = i; j < nligne; j++) {
        if (isunit[j]) continue;
        if ((double)size[j] < sm) {
          sm = size[j];
          pv = j;
        }
      }
      if (pv != i) {
        std::swap(size[pv], size[i]);
        std::swap(ineq[pv], ineq[i]);
      }
    }
    // the reference writes pos[ineq[i]] for unit rows too, whose ineq[i] it never set
    // (traiter.c:577-578 vs 617-618); zero-filled, as our oracle and the reference's fixtures have it
    for (int i = nvar; i < nligne; i++)
      if (ineq[i] >= 0 && ineq[i] < ni) pos[ineq[i]] = i;
    return pos;
  }
  // solution_dual, traiter.c:274-294
  void emit_dual(const HostJob &job, int nvar, int ni, const std::vector<int> &pos) {
    Snap s = download(job);
    push(S_LIST, (E)ni, 0);
    for (int i = 0; i < ni; i++) {
      push(S_FORM, 1, 0);
      const int k = pos[i];
      if (s.flag[k] & PIPAMD_F_UNIT) {
        const int u = s.ref[k];
        E v;
        if (s.flag[0] & PIPAMD_F_UNIT)
          v = (s.ref[0] == u) ? s.den[0] : 0;
        else
          v = s.row(0)[u];
        push(S_VAL, v, s.den[0]);
      } else
        push(S_VAL, 0, 1);
    }
  }

  // ------------------------------------------------------------- traiter()
  void node(HostJob &job, Ctx ctx /* this call's own copy, traiter.c:654 */, int nvar, int nparm, int ni,
            int bigparm, int flags) {
    std::vector<int> pos;
    if (flags & PIPAMD_T_DUAL) pos = dual_positions(job, nvar, ni);
    const int ni0 = ni;
    for (;;) {
      job.pj.status = PIPAMD_ST_RUN;
      std::vector<HostJob *> js{&job};
      const int piv0 = job.pj.npiv;
      run(js);
      pivots += job.pj.npiv - piv0;
      ni = job.pj.ni;
      switch (job.pj.status) {
        case PIPAMD_ST_SOLUTION:
          emit_solution(job, nvar, nparm);
          if (flags & PIPAMD_T_DUAL) emit_dual(job, nvar, ni0, pos);
          return;
        case PIPAMD_ST_NIL: push(S_NIL, 0, 0); return;
        case PIPAMD_ST_CAPACITY:
          grow(job, next_rows(nvar, job.pj.S, job.pj.W), job.pj.W);
          continue;
        case PIPAMD_ST_NEED_PARMCUT:
          if (!host_cut(job, ctx, nvar, nparm, ni, bigparm)) {
            push(S_NIL, 0, 0);
            return;
          }
          continue;
        case PIPAMD_ST_NEED_COMPA: break;
        default: fail(job.pj.status);
      }
      // compa_test, then chercher(Minus) / Critic / Unknown on the refreshed flags
      compa(job, ctx, nvar, nparm, ni);
      Snap s = download(job);
      const int nligne = nvar + ni;
      int pivi = nligne;
      for (int i = 0; i < nligne; i++)
        if (s.flag[i] & PIPAMD_F_MINUS) {
          pivi = i;
          break;
        }
      if (pivi < nligne) continue;  // the kernel pivots on it
      for (int i = 0; i < nligne; i++)
        if (s.flag[i] & PIPAMD_F_CRITIC) {
          pivi = i;
          break;
        }
      if (pivi >= nligne)
        for (int i = 0; i < nligne; i++)
          if (s.flag[i] & PIPAMD_F_UNKNOWN) {
            pivi = i;
            break;
          }
      if (pivi >= nligne) continue;  // all signs settled: the kernel goes on to the solution / cuts
      // ---- the quast forks on the sign of row pivi (traiter.c:695-759)
      if (nparm >= PIPAMD_MAXPARM) fail(PIPAMD_ST_INTERNAL);
      const size_t mark = top_;
      HostJob child = alloc_job(nvar, nparm, ni, bigparm, flags, job.pj.S, job.pj.W);
      if (child.pj.L != job.pj.L || child.pj.S != job.pj.S || child.pj.W != job.pj.W) fail(PIPAMD_ST_INTERNAL);
      copy(d_arena_ + child.block_off, d_arena_ + job.block_off, tab_words(job.pj.L, job.pj.S, job.pj.W) * sizeof(i64),
           hipMemcpyDeviceToDevice);
      child.pj.ldet = job.pj.ldet;
      memcpy(child.pj.det, job.pj.det, sizeof job.pj.det);
      push(S_IF, 0, 0);
      push(S_FORM, (E)(nparm + 1), 0);
      const E *r = s.row(pivi);
      E g = 0;
      for (int j = 0; j < nparm; j++) g = gcd(g, r[j + nvar + 1]);
      if (!(flags & PIPAMD_T_INT)) g = gcd(g, r[nvar]);
      const int nc = ctx.nc;
      ctx.reserve(nc + 1, nparm + 2);
      for (int j = 0; j < nparm; j++) {
        ctx.at(nc, j) = cquo(r[j + nvar + 1], g);
        push(S_VAL, ctx.at(nc, j), 1);
      }
      ctx.at(nc, nparm) = (flags & PIPAMD_T_INT) ? floordiv(r[nvar], g) : cquo(r[nvar], g);
      push(S_VAL, ctx.at(nc, nparm), 1);
      set_flag(child, pivi, PIPAMD_F_PLUS);
      {
        Ctx cc = ctx;
        cc.nc = nc + 1;
        node(child, cc, nvar, nparm, ni, bigparm, flags);
      }
      top_ = mark;
      for (int j = 0; j < nparm; j++) ctx.at(nc, j) = wneg(ctx.at(nc, j));
      ctx.at(nc, nparm) = wneg(wadd(ctx.at(nc, nparm), (E)1));
      ctx.nc = nc + 1;
      set_flag(job, pivi, PIPAMD_F_MINUS);
    }
  }
};
typedef TreeT<i64> Tree;

// tab_simplify (tab.c:396-427) on a row-major matrix
void simplify_rows(i64 *m, int rows, int width, int cst) {
  for (int i = 0; i < rows; i++) {
    i64 *r = m + (size_t)i * width;
    i64 g = 0;
    for (int j = 0; j < width; j++) {
      if (j == cst) continue;
      g = gcd(g, r[j]);
      if (g == 1) break;
    }
    if (g == 0 || g == 1) continue;
    for (int j = 0; j < width; j++) r[j] = (j == cst) ? floordiv(r[j], g) : cquo(r[j], g);
  }
}
void simplify_rows(std::vector<i64> &m, int rows, int width, int cst) { simplify_rows(m.data(), rows, width, cst); }

}  // namespace

// One small 64-bit problem through the device-resident traiter() (defined with the device tree at the end
// of this file): true when it was served there (tape / is_void / pivots filled), false: use the host tree.
// (E: the entry type -- long long, or __int128 for the overflow-safe flavour: the kernel is a template over it)
template <class E>
static bool device_tree_one(pipamd_engine *e, const pipamd_problem &p, int simplify, int deepest_cut, int qflags,
                            std::vector<CellT<E>> &tape, bool *is_void, int64_t *pivots);
// The same for many problems: fills the outputs of every problem it serves and marks it in `served`.
template <class E, class CELL>
static void device_tree_many(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut,
                             std::vector<char> &served, CELL **cells, size_t *n_cells, int *rcs, int *statuses,
                             int64_t *pivots);

// the tape as the C ABI hands it out: (kind, param1, param2) cells, sol.c:52-59
static int export_tape(const std::vector<Cell> &tape, pipamd_sol_cell **cells, size_t *n_cells) {
  *n_cells = tape.size();
  *cells = nullptr;
  if (tape.empty()) return PIPAMD_OK;
  pipamd_sol_cell *c = (pipamd_sol_cell *)malloc(tape.size() * sizeof *c);
  if (!c) return PIPAMD_E_NOMEM;
  for (size_t i = 0; i < tape.size(); i++) {
    c[i].kind = tape[i].kind;
    c[i].reserved = 0;
    c[i].param1 = tape[i].a;
    c[i].param2 = tape[i].b;
  }
  *cells = c;
  return PIPAMD_OK;
}

static int export_tape(const std::vector<CellT<i128>> &tape, pipamd_sol_cell128 **cells, size_t *n_cells) {
  *n_cells = tape.size();
  *cells = nullptr;
  if (tape.empty()) return PIPAMD_OK;
  pipamd_sol_cell128 *c = (pipamd_sol_cell128 *)malloc(tape.size() * sizeof *c);
  if (!c) return PIPAMD_E_NOMEM;
  for (size_t i = 0; i < tape.size(); i++) {
    c[i].kind = tape[i].kind;
    c[i].reserved = 0;
    c[i].param1_lo = (int64_t)(u64)(unsigned __int128)tape[i].a;
    c[i].param1_hi = (int64_t)(tape[i].a >> 64);
    c[i].param2_lo = (int64_t)(u64)(unsigned __int128)tape[i].b;
    c[i].param2_hi = (int64_t)(tape[i].b >> 64);
  }
  *cells = c;
  return PIPAMD_OK;
}

static bool valid_shape(int nvar, int nparm, int ni, int nc, int bigparm, const void *ineq, const void *ctx) {
  if (nvar < 0 || nparm < 0 || ni < 0 || nc < 0 || (ni && !ineq) || (nc && !ctx)) {
    pipamd_set_error("invalid tableau shape or missing rows");
    return false;
  }
  const int ncol = nvar + nparm + 1;
  if (bigparm >= ncol || (bigparm >= 0 && bigparm <= nvar)) {
    pipamd_set_error("bigparm must be -1 or a parameter column (nvar < bigparm < nvar+nparm+1)");
    return false;
  }
  return true;
}

// one problem on an existing tree (device buffers and stream are reused between problems);
// an empty tape with PIPAMD_OK is the front ends' "void" (empty context)
template <class TREE, class CELL>
static int solve_one(TREE &t, int nvar, int nparm, int ni, int nc, int bigparm, int nq, const int64_t *ineq,
                     const int64_t *ctx, int simplify, int deepest_cut, CELL **cells, size_t *n_cells,
                     int *status, int64_t *pivots) {
  if (!cells || !n_cells || !valid_shape(nvar, nparm, ni, nc, bigparm, ineq, ctx)) return PIPAMD_E_INVALID;
  const int ncol = nvar + nparm + 1;
  *cells = nullptr;
  *n_cells = 0;
  if (status) *status = 0;
  if (pivots) *pivots = 0;
  std::vector<i64> a((const i64 *)ineq, (const i64 *)ineq + (size_t)ni * ncol);
  std::vector<i64> c((const i64 *)ctx, (const i64 *)ctx + (size_t)nc * (nparm + 1));
  if (nq && simplify) {  // maind.c:190-196
    simplify_rows(a, ni, ncol, nvar);
    simplify_rows(c, nc, nparm + 1, nparm);
  }
  t.reset(deepest_cut);
  int rc = PIPAMD_OK;
  bool non_void = false;
  try {
    non_void = t.front(nvar, nparm, ni, nc, bigparm, nq, a.data(), c.data());
  } catch (int code) {
    rc = code;
    if (status) *status = t.fail_status;
    if (rc == PIPAMD_E_SOLVER)
      pipamd_set_error("solver stopped with status %d (5 = the reference's \"Integer overflow\" exit)", t.fail_status);
  }
  if (pivots) *pivots = t.pivots;
  if (rc) return rc;
  return non_void ? export_tape(t.tape, cells, n_cells) : PIPAMD_OK;
}

extern "C" int pipamd_solve_tableau(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int nq,
                                    const int64_t *ineq, const int64_t *ctx, int simplify, int deepest_cut,
                                    pipamd_sol_cell **cells, size_t *n_cells, int *status, int64_t *pivots) {
  if (!e) return PIPAMD_E_INVALID;
  if (hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;
  try {
    if (cells && n_cells && valid_shape(nvar, nparm, ni, nc, bigparm, ineq, ctx)) {  // a small problem: wholly on the device
      const pipamd_problem p{nvar, nparm, ni, nc, bigparm, nq, ineq, ctx};
      std::vector<Cell> tape;
      bool is_void = false;
      int64_t pv = 0;
      if (device_tree_one<i64>(e, p, simplify, deepest_cut, 0, tape, &is_void, &pv)) {
        *cells = nullptr;
        *n_cells = 0;
        if (status) *status = 0;
        if (pivots) *pivots = pv;
        return is_void ? PIPAMD_OK : export_tape(tape, cells, n_cells);
      }
    }
    Tree t(e, deepest_cut);
    return solve_one(t, nvar, nparm, ni, nc, bigparm, nq, ineq, ctx, simplify, deepest_cut, cells, n_cells, status, pivots);
  } catch (int code) {
    return code;
  }
}

// traiter() (traiter.c:628-791) behind the reference's own front ends: see include/piplib_amd.h
template <class E, class CELL>
static int traiter_any(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int flags, int deepest_cut,
                       const int64_t *tableau, const int64_t *context, CELL **cells, size_t *n_cells, int *status,
                       int64_t *pivots) {
  if (!e || !cells || !n_cells || !valid_shape(nvar, nparm, ni, nc, bigparm, tableau, context)) return PIPAMD_E_INVALID;
  if ((flags & ~(PIPAMD_T_INT | PIPAMD_T_DUAL)) || ((flags & PIPAMD_T_INT) && (flags & PIPAMD_T_DUAL))) {
    pipamd_set_error("pipamd_traiter: flags must be 0, PIPAMD_T_INT or PIPAMD_T_DUAL (the dual needs a rational solve)");
    return PIPAMD_E_INVALID;
  }
  if (hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;
  *cells = nullptr;
  *n_cells = 0;
  if (status) *status = 0;
  if (pivots) *pivots = 0;
  int rc = PIPAMD_OK;
  try {
    {  // a small problem (with or without the dual): wholly on the device
      const pipamd_problem p{nvar, nparm, ni, nc, bigparm, (flags & PIPAMD_T_INT) ? 1 : 0, tableau, context};
      std::vector<CellT<E>> tape;
      bool is_void = false;
      int64_t pv = 0;
      if (device_tree_one<E>(e, p, 0, deepest_cut, Q_NO_CONTEXT_TEST | ((flags & PIPAMD_T_DUAL) ? Q_DUAL : 0), tape, &is_void, &pv)) {
        if (pivots) *pivots = pv;
        return export_tape(tape, cells, n_cells);
      }
    }
    TreeT<E> t(e, deepest_cut);
    try {
      t.traiter_call(nvar, nparm, ni, nc, bigparm, flags, (const i64 *)tableau, (const i64 *)context);
    } catch (int code) {
      rc = code;
      if (status) *status = t.fail_status;
      if (rc == PIPAMD_E_SOLVER)
        pipamd_set_error("solver stopped with status %d (5 = the reference's \"Integer overflow\" exit)", t.fail_status);
    }
    if (pivots) *pivots = t.pivots;
    if (!rc) rc = export_tape(t.tape, cells, n_cells);
  } catch (int code) {
    rc = code;
  }
  return rc;
}
extern "C" int pipamd_traiter(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int flags,
                              int deepest_cut, const int64_t *tableau, const int64_t *context,
                              pipamd_sol_cell **cells, size_t *n_cells, int *status, int64_t *pivots) {
  return traiter_any<i64>(e, nvar, nparm, ni, nc, bigparm, flags, deepest_cut, tableau, context, cells, n_cells, status,
                          pivots);
}
// the overflow-safe flavour (include/piplib/piplib.h:42-88): 128-bit entries on the device and in
// the host tree, inputs still int64, cells with 128-bit parameters
extern "C" int pipamd_traiter128(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int flags,
                                 int deepest_cut, const int64_t *tableau, const int64_t *context,
                                 pipamd_sol_cell128 **cells, size_t *n_cells, int *status, int64_t *pivots) {
  return traiter_any<i128>(e, nvar, nparm, ni, nc, bigparm, flags, deepest_cut, tableau, context, cells, n_cells, status,
                           pivots);
}
extern "C" int pipamd_solve_tableau128(pipamd_engine *e, int nvar, int nparm, int ni, int nc, int bigparm, int nq,
                                       const int64_t *ineq, const int64_t *ctx, int simplify, int deepest_cut,
                                       pipamd_sol_cell128 **cells, size_t *n_cells, int *status, int64_t *pivots) {
  if (!e) return PIPAMD_E_INVALID;
  if (hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;
  try {
    if (cells && n_cells && valid_shape(nvar, nparm, ni, nc, bigparm, ineq, ctx)) {  // a small problem: wholly on the device
      const pipamd_problem p{nvar, nparm, ni, nc, bigparm, nq, ineq, ctx};
      std::vector<CellT<i128>> tape;
      bool is_void = false;
      int64_t pv = 0;
      if (device_tree_one<i128>(e, p, simplify, deepest_cut, 0, tape, &is_void, &pv)) {
        *cells = nullptr;
        *n_cells = 0;
        if (status) *status = 0;
        if (pivots) *pivots = pv;
        return is_void ? PIPAMD_OK : export_tape(tape, cells, n_cells);
      }
    }
    TreeT<i128> t(e, deepest_cut);
    return solve_one(t, nvar, nparm, ni, nc, bigparm, nq, ineq, ctx, simplify, deepest_cut, cells, n_cells, status, pivots);
  } catch (int code) {
    return code;
  }
}

// Many independent problems: `nthreads` host threads, each with its own tree (device arena and
// HIP stream), pull problems from a shared counter; their launches overlap on the GPU.
// rc[i] receives what pipamd_solve_tableau would have returned for problem i.
// TREE = Tree (64-bit entries; small problems go to the device-resident traiter() first) or TreeT<i128>.
template <class TREE, class CELL>
static int solve_many(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut, int nthreads,
                      CELL **cells, size_t *n_cells, int *rcs, int *statuses, int64_t *pivots, std::vector<char> &served) {
  std::atomic<int> next(0);
  const int device = e->device;
  auto worker = [&]() {
    if (hipSetDevice(device) != hipSuccess) return;
    try {
      TREE t(e, deepest_cut);
      for (;;) {
        const int i = next.fetch_add(1);
        if (i >= n) break;
        if (served[i]) continue;
        const pipamd_problem &p = probs[i];
        int st = 0;
        int64_t pv = 0;
        rcs[i] = solve_one(t, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx, simplify, deepest_cut,
                           &cells[i], &n_cells[i], &st, &pv);
        if (statuses) statuses[i] = st;
        if (pivots) pivots[i] = pv;
      }
    } catch (int) {  // the tree could not be set up (stream creation failed): its problems stay E_HIP
    }
  };
  std::vector<std::thread> th;
  for (int k = 0; k < nthreads; k++) th.emplace_back(worker);
  for (auto &x : th) x.join();
  return PIPAMD_OK;
}

template <class CELL>
static bool many_args(pipamd_engine *e, int n, const pipamd_problem *probs, int &nthreads, CELL **cells, size_t *n_cells,
                      int *rcs, int *statuses, int64_t *pivots) {
  if (!e || n < 0 || (n && (!probs || !cells || !n_cells || !rcs))) return false;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > n) nthreads = n > 0 ? n : 1;
  for (int i = 0; i < n; i++) {  // a worker that cannot even start leaves its problems marked as failed
    rcs[i] = PIPAMD_E_HIP;
    cells[i] = nullptr;
    n_cells[i] = 0;
    if (statuses) statuses[i] = 0;
    if (pivots) pivots[i] = 0;
  }
  return true;
}

extern "C" int pipamd_solve_tableaux(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut,
                                     int nthreads, pipamd_sol_cell **cells, size_t *n_cells, int *rcs, int *statuses,
                                     int64_t *pivots) {
  if (!many_args(e, n, probs, nthreads, cells, n_cells, rcs, statuses, pivots)) return PIPAMD_E_INVALID;
  std::vector<char> served(n > 0 ? n : 1, 0);
  if (hipSetDevice(e->device) == hipSuccess)  // small problems: wholly on the device, in one launch
    device_tree_many<i64>(e, n, probs, simplify, deepest_cut, served, cells, n_cells, rcs, statuses, pivots);
  return solve_many<Tree, pipamd_sol_cell>(e, n, probs, simplify, deepest_cut, nthreads, cells, n_cells, rcs, statuses, pivots,
                                           served);
}

// The same on 128-bit entries (the overflow-safe flavour): one TreeT<__int128> per host thread.
extern "C" int pipamd_solve_tableaux128(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut,
                                        int nthreads, pipamd_sol_cell128 **cells, size_t *n_cells, int *rcs, int *statuses,
                                        int64_t *pivots) {
  if (!many_args(e, n, probs, nthreads, cells, n_cells, rcs, statuses, pivots)) return PIPAMD_E_INVALID;
  std::vector<char> served(n > 0 ? n : 1, 0);
  if (hipSetDevice(e->device) == hipSuccess)  // small problems: wholly on the device, in one launch
    device_tree_many<i128>(e, n, probs, simplify, deepest_cut, served, cells, n_cells, rcs, statuses, pivots);
  return solve_many<TreeT<i128>, pipamd_sol_cell128>(e, n, probs, simplify, deepest_cut, nthreads, cells, n_cells, rcs, statuses,
                                                     pivots, served);
}

// =========================================================================== Forest
// Lock-step scheduler for MANY parametric problems.  The per-problem Tree above pays a few host
// <-> device round trips per decision; on tiny problems that latency dominates.  The Forest
// keeps one explicit frame stack per problem (the same traiter() state machine, no recursion)
// and serves every problem of the batch with ONE sequence per step:
//   clone pass (tree splits) -> patch pass (new tableaux, flags, cut rows) -> advance launch
//   (all runnable jobs of all problems) -> gather pass (undecided rows / cut rows / solutions).
// Problems that hit a rare path (capacity growth, deepest cuts, dual) are handed to the Tree.
namespace {

template <class E> struct FResultT {
  int rc = PIPAMD_OK, status = 0;
  long long pivots = 0;
  bool is_void = false;
  std::vector<CellT<E>> tape;
};
typedef FResultT<i64> FResult;

// E = the entry type of the device tableaux, of the contexts, cut rows and tape cells on the host: long long, or __int128
// for the overflow-safe flavour (the same scheduler; blocks, patches and gathered records are EW = 1 or 2 int64 words a value).
template <class E>
class ForestT {
  typedef CtxT<E> Ctx;
  typedef CellT<E> Cell;
  typedef FResultT<E> FResult;
  static constexpr int EW = (int)(sizeof(E) / 8), EBITS = 64 * EW;

 public:
  explicit ForestT(int device) {
    (void)device;
    const hipError_t err = hipStreamCreateWithFlags(&st_, hipStreamNonBlocking);
    if (err != hipSuccess) {
      st_ = 0;
      pipamd_set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(err));
      throw (int)PIPAMD_E_HIP;
    }
  }
  ~ForestT() {
    void *bufs[] = {d_arena_, d_jobs_, d_off_, d_out_, d_patch_, d_pidx_, d_clone_, d_fresh_, d_fidx_, d_big_};
    for (void *b : bufs)
      if (b) hipFree(b);
    if (st_) hipStreamDestroy(st_);
  }

  // returns per-problem results; results[i].rc == PIPAMD_E_TOOLARGE marks "use the Tree path"
  void solve(int n, const pipamd_problem *probs, int simplify, std::vector<FResult> &res) {
    res.assign(n, FResult());
    P_.assign(n, Prob());
    jobs_.clear();
    is_sub_.clear();
    // ---- regions
    size_t total = 0;
    for (int i = 0; i < n; i++) {
      const pipamd_problem &p = probs[i];
      Prob &q = P_[i];
      q.nvar = p.nvar;
      q.nparm0 = p.nparm;
      q.region_words = region_words(p);
      q.region_off = total;
      q.top = 0;
      total += q.region_words;
    }
    if (total * sizeof(i64) > arena_cap_) {  // exact size: the arena is the one large buffer (no doubling)
      if (d_arena_) hipFree(d_arena_);
      d_arena_ = nullptr;
      arena_cap_ = 0;
      HIPTHROW(hipMalloc((void **)&d_arena_, total * sizeof(i64)));
      arena_cap_ = total * sizeof(i64);
    }
    // ---- start every problem
    for (int i = 0; i < n; i++) {
      try {
        start(i, probs[i], simplify);
      } catch (int code) {
        finish_error(i, code);
      }
    }
    // ---- lock-step loop
    const bool stats = getenv("PIPAMD_FOREST_STATS") != nullptr;
    double t_dev = 0, t_host = 0;
    long long n_launched = 0, n_patchwords = 0;
    int nsteps = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int step = 0; step < 200000; step++) {
      std::vector<int> lj;
      for (int i = 0; i < n; i++) {
        Prob &q = P_[i];
        if (q.done) continue;
        Frame &f = q.stack.back();
        if (f.phase == PH_RUN)
          lj.push_back(f.job);
        else
          for (int k = 0; k < f.sub_count; k++)
            if (jobs_[f.sub_begin + k].status == PIPAMD_ST_RUN) lj.push_back(f.sub_begin + k);
      }
      if (lj.empty()) break;
      const double t0 = now();
      n_launched += (long long)lj.size();
      n_patchwords += (long long)patch_.size() + 2 * (long long)fresh_.size();
      step_device(lj);
      const double t1 = now();
      for (int i = 0; i < n; i++) {
        if (P_[i].done) continue;
        try {
          advance(i);
        } catch (int code) {
          finish_error(i, code);
        }
      }
      t_dev += t1 - t0;
      t_host += now() - t1;
      nsteps++;
    }
    if (stats)
      fprintf(stderr, "[forest] %d problems, %d steps, %lld job launches, %.1f MB patches; device steps %.1f ms, host %.1f ms\n",
              n, nsteps, n_launched, n_patchwords * 4e-6, t_dev * 1e3, t_host * 1e3);
    for (int i = 0; i < n; i++) {
      Prob &q = P_[i];
      if (!q.done) {
        q.rc = PIPAMD_E_SOLVER;
        q.status = PIPAMD_ST_INTERNAL;
      }
      res[i].rc = q.rc;
      res[i].status = q.status;
      res[i].pivots = q.pivots;
      res[i].is_void = q.is_void;
      res[i].tape.swap(q.tape);
    }
  }

 private:
  enum { PH_RUN = 0, PH_COMPA = 1 };
  enum { K_CTX = 0, K_NODE = 1 };
  struct Frame {
    int kind = K_NODE, phase = PH_RUN;
    int job = -1;
    Ctx ctx;
    int nparm = 0, ni = 0, bigparm = -1, flags = 0;
    int last_npiv = 0;
    size_t mark = 0;  // region top when the frame was pushed (released when it pops)
    // compa_test in flight
    std::vector<int> rows, critic;
    std::vector<std::vector<E>> rowvals;  // constant | parameters of each undecided row
    int sub_begin = 0, sub_count = 0;
    // split in flight (this frame is the parent waiting for its "then" child)
    int split_row = -1;
  };
  struct Prob {
    int nvar = 0, nparm0 = 0;
    size_t region_off = 0, region_words = 0, top = 0;
    std::vector<Frame> stack;
    std::vector<Cell> tape;
    long long pivots = 0;
    int rc = PIPAMD_OK, status = 0;
    bool done = false, is_void = false;
    // pending main problem (started after the context test)
    std::vector<i64> ineq;
    int ni = 0, bigparm = -1, nq = 1;
  };

  hipStream_t st_ = 0;
  std::vector<Prob> P_;
  std::vector<PipJob> jobs_;  // master copies, index = job id
  std::vector<char> is_sub_;  // job id -> compa_test / context sub-problem (only its status matters)
  std::vector<int> lpos_;     // job id -> position in the last launch (or -1)
  std::vector<i64> gout_, goff_;
  // device buffers
  i64 *d_arena_ = nullptr, *d_off_ = nullptr, *d_out_ = nullptr, *d_pidx_ = nullptr, *d_clone_ = nullptr;
  i64 *d_fresh_ = nullptr, *d_fidx_ = nullptr;
  size_t fresh_cap_ = 0, fidx_cap_ = 0;
  PipJob *d_jobs_ = nullptr;
  // a launch whose combined row tables outgrow LDS (a tall and a wide job in one step) keeps them in HBM
  void *d_big_ = nullptr;
  size_t big_bytes_ = 0;
  void *big_[2] = {&d_big_, &big_bytes_};
  int *d_patch_ = nullptr;
  size_t arena_cap_ = 0, jobs_cap_ = 0, off_cap_ = 0, out_cap_ = 0, patch_cap_ = 0, pidx_cap_ = 0, clone_cap_ = 0;
  // staging of the current step
  std::vector<int> patch_;
  std::vector<i64> pidx_, clones_, fresh_, fidx_;

  static int even(int x) { return (x + 1) & ~1; }
 public:
  // device words reserved for one problem: 24 tableaux of the main shape (tree depth) and the
  // sub-problems of one compa_test round
  static size_t region_words(const pipamd_problem &p) {
    const size_t mainb = block_words(p.nvar, p.ni + 24, p.nvar + p.nparm + 1 + 6);
    const size_t subb = block_words(p.nparm + 8, p.nc + 48, p.nparm + 8 + 1);
    return 24 * mainb + (size_t)(2 * (p.nvar + p.ni + 24) + 4) * subb;
  }

 private:
  static size_t block_words(int nvar, int S, int W) {
    W = even(W);
    const int L = even(nvar + S);
    return (size_t)L * EW + (size_t)L + (size_t)S * W * EW + (size_t)even(nvar * (W - nvar) + nvar) * EW +
           (size_t)even(S * 8 + (3 * L + 7) / 8) + 2 * PIPAMD_DETLOG * EW;
  }
  template <class T>
  void ensure(T *&buf, size_t &cap, size_t bytes) {
    if (bytes <= cap) return;
    if (buf) hipFree(buf);
    cap = bytes * 2 + 4096;
    HIPTHROW(hipMalloc((void **)&buf, cap));
  }
  void fail(int i, int st) {
    P_[i].status = st;
    throw (int)PIPAMD_E_SOLVER;
  }
  void finish_error(int i, int code) {
    Prob &q = P_[i];
    q.done = true;
    q.rc = code;
  }
  void tape_push(int i, int kind, E a, E b) {
    P_[i].tape.push_back(Cell{kind, a, b});
    if (P_[i].tape.size() >= 4096) fail(i, PIPAMD_ST_INTERNAL);
  }

  // ---- patches ------------------------------------------------------------
  void patch32(size_t dst32, const int *data, int n32) {
    pidx_.push_back((i64)patch_.size());
    patch_.push_back((int)(unsigned)(dst32 & 0xffffffffu));
    patch_.push_back((int)(dst32 >> 32));
    patch_.push_back(n32);
    patch_.insert(patch_.end(), data, data + n32);
  }
  // n values of the entry type at int64 word dst64
  void patch_vals(size_t dst64, const E *data, size_t n) { patch32(dst64 * 2, (const int *)data, (int)(n * 2 * EW)); }
  void patch_flag(const PipJob &pj, int row, int f) { patch32(((size_t)pj.rows_off + (size_t)pj.L * EW) * 2 + row, &f, 1); }

  // ---- jobs ---------------------------------------------------------------
  // allocate a job block in problem i's region; throws TOOLARGE (-> Tree path) when it is full
  int new_job(int i, int nvar, int nparm, int ni, int bigparm, int tflags, int S, int W) {
    Prob &q = P_[i];
    PipJob pj;
    memset(&pj, 0, sizeof pj);
    S = std::max(S, ni + 1);
    W = even(std::max(W, nvar + nparm + 1));
    if (W > PIPAMD_MAXCOL || S > PIPAMD_SMAX || nvar + S > PIPAMD_LMAX ||
        pipk_advance_lds_bytes((even(nvar + S) + 3) & ~3, (S + 3) & ~3, W, EBITS) > PIPAMD_LDS_BUDGET)
      throw (int)PIPAMD_E_TOOLARGE;
    const int L = even(nvar + S);
    const size_t words = block_words(nvar, S, W);
    if (q.top + words > q.region_words) throw (int)PIPAMD_E_TOOLARGE;
    const i64 off = (i64)(q.region_off + q.top);
    q.top += words;
    pj.rows_off = off;
    pj.vals_off = off + (i64)L * EW + (i64)L;
    pj.sol_off = pj.vals_off + (i64)S * W * EW;
    pj.state_off = pj.sol_off + (i64)even(nvar * (W - nvar) + nvar) * EW;
    pj.log_off = off + (i64)words - 2 * PIPAMD_DETLOG * EW;
    pj.nvar = nvar;
    pj.nparm = nparm;
    pj.ni = ni;
    pj.bigparm = bigparm;
    pj.tflags = tflags | PIPAMD_T_SORT;
    pj.L = L;
    pj.S = S;
    pj.W = W;
    pj.status = PIPAMD_ST_RUN;
    pj.ldet = 1;
    pj.det[0] = 1;
    pj.ebits = EBITS;
    jobs_.push_back(pj);
    is_sub_.push_back(0);
    return (int)jobs_.size() - 1;
  }
  // tab_alloc + tab_get (tab.c:158-248): only the rows travel, a kernel builds the block
  void fresh_begin(int job) {
    const PipJob &pj = jobs_[job];
    fidx_.push_back((i64)fresh_.size());
    const i64 hdr[8] = {pj.rows_off, pj.nvar, pj.ni, pj.nvar + pj.nparm + 1, pj.L, pj.S, pj.W, 0};
    fresh_.insert(fresh_.end(), hdr, hdr + 8);
  }
  // values of the entry type behind a record's header (EW words each, low word first)
  void fresh_vals(const E *v, size_t n) {
    const i64 *w = (const i64 *)v;
    fresh_.insert(fresh_.end(), w, w + n * EW);
  }
  void fresh_block(int job, const std::vector<i64> &rows) {  // (input rows: long longs in either flavour)
    fresh_begin(job);
    for (i64 x : rows) {
      const E v = (E)x;
      fresh_vals(&v, 1);
    }
  }
  int context_job(int i, const Ctx &ctx, int nparm, int nc, const std::vector<E> *extra) {
    const int ni = nc + (extra ? 1 : 0), ncol = nparm + 1;
    const int job = new_job(i, nparm, 0, ni, -1, PIPAMD_T_INT, ni + 16, ncol);
    fresh_begin(job);
    for (int k = 0; k < nc; k++) fresh_vals(&ctx.v[(size_t)k * ctx.width], ncol);
    if (extra) fresh_vals(extra->data(), ncol);
    is_sub_[job] = 1;
    return job;
  }

  void start(int i, const pipamd_problem &p, int simplify) {
    Prob &q = P_[i];
    const int ncol = p.nvar + p.nparm + 1;
    if (p.nvar < 0 || p.nparm < 0 || p.ni < 0 || p.nc < 0 || (p.ni && !p.ineq) || (p.nc && !p.ctx) || p.bigparm >= ncol ||
        (p.bigparm >= 0 && p.bigparm <= p.nvar)) {
      q.done = true;
      q.rc = PIPAMD_E_INVALID;
      return;
    }
    q.ineq.assign((const i64 *)p.ineq, (const i64 *)p.ineq + (size_t)p.ni * ncol);
    std::vector<i64> c((const i64 *)p.ctx, (const i64 *)p.ctx + (size_t)p.nc * (p.nparm + 1));
    if (p.nq && simplify) {
      simplify_rows(q.ineq, p.ni, ncol, p.nvar);
      simplify_rows(c, p.nc, p.nparm + 1, p.nparm);
    }
    q.ni = p.ni;
    q.bigparm = p.bigparm;
    q.nq = p.nq;
    Frame f;
    f.ctx.reserve(p.nc + 4, p.nparm + 2);
    f.ctx.nc = p.nc;
    for (int r = 0; r < p.nc; r++)
      for (int k = 0; k <= p.nparm; k++) f.ctx.at(r, k) = (E)c[(size_t)r * (p.nparm + 1) + k];
    f.nparm = p.nparm;
    f.mark = 0;
    if (p.nc) {  // context emptiness test first (maind.c:196-203)
      f.kind = K_CTX;
      f.job = context_job(i, f.ctx, p.nparm, p.nc, nullptr);
      q.stack.push_back(f);
    } else {
      q.stack.push_back(f);
      begin_main(i);
    }
  }
  // replace the (finished) context frame by the main traiter() frame
  void begin_main(int i) {
    Prob &q = P_[i];
    Frame &f = q.stack.back();
    const int tfl = q.nq ? PIPAMD_T_INT : 0;
    q.top = f.mark;
    f.kind = K_NODE;
    f.phase = PH_RUN;
    f.ni = q.ni;
    f.bigparm = q.bigparm;
    f.flags = tfl;
    f.job = new_job(i, q.nvar, f.nparm, q.ni, q.bigparm, tfl, q.ni + 24, q.nvar + f.nparm + 1 + (f.nparm ? 6 : 0));
    f.mark = 0;
    f.last_npiv = 0;
    fresh_block(f.job, q.ineq);
  }

  // ---- one device step ---------------------------------------------------------------
  void step_device(const std::vector<int> &lj) {
    const int n = (int)lj.size();
    std::vector<PipJob> tab(n);
    lpos_.assign(jobs_.size(), -1);
    goff_.assign(n + 1, 0);
    int Lm = 4, Sm = 4, Wm = 2;
    for (int k = 0; k < n; k++) {
      tab[k] = jobs_[lj[k]];
      lpos_[lj[k]] = k;
      Lm = std::max(Lm, (int)tab[k].L);
      Sm = std::max(Sm, (int)tab[k].S);
      Wm = std::max(Wm, (int)tab[k].W);
    }
    ensure(d_jobs_, jobs_cap_, sizeof(PipJob) * n);
    ensure(d_off_, off_cap_, sizeof(i64) * (n + 1));
    if (!clones_.empty()) {
      ensure(d_clone_, clone_cap_, sizeof(i64) * clones_.size());
      HIPTHROW(hipMemcpyAsync(d_clone_, clones_.data(), sizeof(i64) * clones_.size(), hipMemcpyHostToDevice, st_));
      HIPTHROW(pipk_launch_clone(d_arena_, d_clone_, (int)(clones_.size() / 3), st_));
    }
    if (!fidx_.empty()) {  // new tableaux first: patches (flags of a fresh clone, ...) come after
      ensure(d_fresh_, fresh_cap_, sizeof(i64) * fresh_.size());
      ensure(d_fidx_, fidx_cap_, sizeof(i64) * fidx_.size());
      HIPTHROW(hipMemcpyAsync(d_fresh_, fresh_.data(), sizeof(i64) * fresh_.size(), hipMemcpyHostToDevice, st_));
      HIPTHROW(hipMemcpyAsync(d_fidx_, fidx_.data(), sizeof(i64) * fidx_.size(), hipMemcpyHostToDevice, st_));
      HIPTHROW(pipk_launch_fresh(d_arena_, d_fresh_, d_fidx_, (int)fidx_.size(), EBITS, st_));
    }
    if (!pidx_.empty()) {
      ensure(d_patch_, patch_cap_, sizeof(int) * patch_.size());
      ensure(d_pidx_, pidx_cap_, sizeof(i64) * pidx_.size());
      HIPTHROW(hipMemcpyAsync(d_patch_, patch_.data(), sizeof(int) * patch_.size(), hipMemcpyHostToDevice, st_));
      HIPTHROW(hipMemcpyAsync(d_pidx_, pidx_.data(), sizeof(i64) * pidx_.size(), hipMemcpyHostToDevice, st_));
      HIPTHROW(pipk_launch_patch(d_arena_, d_patch_, d_pidx_, (int)pidx_.size(), st_));
    }
    HIPTHROW(hipMemcpyAsync(d_jobs_, tab.data(), sizeof(PipJob) * n, hipMemcpyHostToDevice, st_));
    // the staging vectors must stay alive until the copies are done: sync once before reuse
    for (int guard = 0; guard < 4096; guard++) {
      HIPTHROW(pipk_launch_advance_q(d_jobs_, d_arena_, n, Lm, Sm, Wm, 1 << 20, n >= 2048 ? 1 : 4, EBITS, nullptr, 0, big_, 0,
                                     nullptr, st_));
      HIPTHROW(hipMemcpyAsync(tab.data(), d_jobs_, sizeof(PipJob) * n, hipMemcpyDeviceToHost, st_));
      HIPTHROW(hipStreamSynchronize(st_));
      bool again = false;
      for (int k = 0; k < n; k++)
        if (tab[k].status == PIPAMD_ST_RUN) again = true;
      if (!again) break;
    }
    clones_.clear();
    patch_.clear();
    pidx_.clear();
    fresh_.clear();
    fidx_.clear();
    // gather only what the host will read: nothing for sub-problems (their status is the answer)
    for (int k = 0; k < n; k++) {
      size_t need = 0;
      if (!is_sub_[lj[k]]) {
        if (tab[k].status == PIPAMD_ST_NEED_COMPA)
          need = 1 + (size_t)(tab[k].nvar + tab[k].ni) * (3 + tab[k].nparm);
        else if (tab[k].status == PIPAMD_ST_NEED_PARMCUT)
          need = 2 + (size_t)tab[k].W;
        else if (tab[k].status == PIPAMD_ST_SOLUTION)
          need = (size_t)tab[k].nvar * (tab[k].nparm + 2);
      }
      goff_[k + 1] = goff_[k] + (i64)need * EW;
    }
    ensure(d_out_, out_cap_, sizeof(i64) * (size_t)goff_[n] + 8);
    HIPTHROW(hipMemcpyAsync(d_off_, goff_.data(), sizeof(i64) * (n + 1), hipMemcpyHostToDevice, st_));
    HIPTHROW(pipk_launch_gather(d_jobs_, d_arena_, n, d_out_, d_off_, EBITS, st_));
    gout_.resize((size_t)goff_[n]);
    if (goff_[n])
      HIPTHROW(hipMemcpyAsync(gout_.data(), d_out_, sizeof(i64) * (size_t)goff_[n], hipMemcpyDeviceToHost, st_));
    HIPTHROW(hipStreamSynchronize(st_));
    for (int k = 0; k < n; k++) jobs_[lj[k]] = tab[k];
  }
  const E *gathered(int job) const { return (const E *)(gout_.data() + goff_[lpos_[job]]); }

  // ---- host side of one problem after a step ----------------------------------------
  void pop_frame(int i) {
    Prob &q = P_[i];
    q.top = q.stack.back().mark;
    q.stack.pop_back();
    if (q.stack.empty()) {
      q.done = true;
      return;
    }
    // the parent was waiting for its "then" branch: now the "else" branch (traiter.c:751-758)
    Frame &f = q.stack.back();
    const int nc = f.ctx.nc;
    for (int j = 0; j < f.nparm; j++) f.ctx.at(nc, j) = wneg(f.ctx.at(nc, j));
    f.ctx.at(nc, f.nparm) = wneg(wadd(f.ctx.at(nc, f.nparm), (E)1));
    f.ctx.nc = nc + 1;
    patch_flag(jobs_[f.job], f.split_row, PIPAMD_F_MINUS);
    jobs_[f.job].status = PIPAMD_ST_RUN;
    f.split_row = -1;
    f.phase = PH_RUN;
  }

  void advance(int i) {
    Prob &q = P_[i];
    Frame &f = q.stack.back();
    if (f.phase == PH_COMPA) {
      finish_compa(i);
      return;
    }
    PipJob &pj = jobs_[f.job];
    q.pivots += pj.npiv - f.last_npiv;
    f.last_npiv = pj.npiv;
    f.ni = pj.ni;
    const int st = pj.status;
    if (f.kind == K_CTX) {
      if (st == PIPAMD_ST_NIL) {
        q.is_void = true;
        q.done = true;
        return;
      }
      if (st != PIPAMD_ST_SOLUTION) {
        if (st == PIPAMD_ST_CAPACITY || st == PIPAMD_ST_NEED_PARMCUT) throw (int)PIPAMD_E_TOOLARGE;
        fail(i, st);
      }
      begin_main(i);
      return;
    }
    switch (st) {
      case PIPAMD_ST_SOLUTION: {
        const E *g = gathered(f.job);
        const int nvar = q.nvar, np = f.nparm;
        const size_t nn = (size_t)nvar * (np + 1);
        tape_push(i, S_LIST, (E)nvar, (E)0);
        for (int r = 0; r < nvar; r++) {
          tape_push(i, S_FORM, (E)(np + 1), (E)0);
          for (int j = 0; j <= np; j++) tape_push(i, S_VAL, g[(size_t)r * (np + 1) + j], g[nn + r]);
        }
        pop_frame(i);
        return;
      }
      case PIPAMD_ST_NIL:
        tape_push(i, S_NIL, (E)0, (E)0);
        pop_frame(i);
        return;
      case PIPAMD_ST_NEED_PARMCUT: parm_cut(i); return;
      case PIPAMD_ST_NEED_COMPA: start_compa(i); return;
      case PIPAMD_ST_CAPACITY: throw (int)PIPAMD_E_TOOLARGE;
      default: fail(i, st);
    }
  }

  // compa_test (traiter.c:162-243): sub-problems of every undecided row, solved next step
  void start_compa(int i) {
    Prob &q = P_[i];
    Frame &f = q.stack.back();
    if (f.nparm >= PIPAMD_MAXPARM) fail(i, PIPAMD_ST_INTERNAL);
    const E *g = gathered(f.job);
    const int nrec = (int)g[0], rec = 3 + f.nparm, np = f.nparm;
    f.rows.clear();
    f.critic.clear();
    f.rowvals.clear();
    f.sub_begin = (int)jobs_.size();
    f.sub_count = 0;
    for (int t = 0; t < nrec; t++) {
      const E *r = g + 1 + (size_t)t * rec;
      f.rows.push_back((int)r[0]);
      f.critic.push_back((int)r[1]);
      std::vector<E> rv(r + 2, r + 3 + np);  // constant | parameters
      f.rowvals.push_back(rv);
      std::vector<E> ex(np + 1);
      for (int j = 0; j < np; j++) ex[j] = rv[1 + j];
      ex[np] = r[1] != 0 ? rv[0] : wsub(rv[0], (E)1);
      context_job(i, f.ctx, np, f.ctx.nc, &ex);
      for (int j = 0; j < np; j++) ex[j] = wneg(rv[1 + j]);
      ex[np] = wsub(wneg(rv[0]), (E)1);
      context_job(i, f.ctx, np, f.ctx.nc, &ex);
      f.sub_count += 2;
    }
    f.phase = PH_COMPA;
    if (nrec == 0) finish_compa(i);
  }

  void finish_compa(int i) {
    Prob &q = P_[i];
    Frame &f = q.stack.back();
    for (int k = 0; k < f.sub_count; k++) {
      const int st = jobs_[f.sub_begin + k].status;
      if (st == PIPAMD_ST_RUN) return;  // not all there yet
    }
    const PipJob &pj = jobs_[f.job];
    const int nrows = (int)f.rows.size();
    std::vector<int> newflag(nrows, -1);
    for (int t = 0; t < nrows; t++) {
      const PipJob &jp = jobs_[f.sub_begin + 2 * t], &jm = jobs_[f.sub_begin + 2 * t + 1];
        q.pivots += jp.npiv + jm.npiv;
      for (const PipJob *s : {&jp, &jm}) {
        if (s->status == PIPAMD_ST_CAPACITY || s->status == PIPAMD_ST_NEED_PARMCUT) throw (int)PIPAMD_E_TOOLARGE;
        if (s->status != PIPAMD_ST_SOLUTION && s->status != PIPAMD_ST_NIL) fail(i, s->status);
      }
      const bool cp = jp.status != PIPAMD_ST_NIL, cm = jm.status != PIPAMD_ST_NIL;
      int fl;
      if (cp && cm)
        fl = f.critic[t] ? PIPAMD_F_CRITIC : PIPAMD_F_UNKNOWN;
      else if (cm)
        fl = PIPAMD_F_MINUS;
      else
        fl = cp ? PIPAMD_F_PLUS : PIPAMD_F_ZERO;
      newflag[t] = fl;
      patch_flag(pj, f.rows[t], fl);
      if (fl == PIPAMD_F_MINUS) break;
    }
    // release the sub-problems (they sit above everything this frame still needs)
    // rows behind the first negative one keep their old flag (Critic or Unknown)
    int pick = -1;
    for (int t = 0; t < nrows && pick < 0; t++)
      if (newflag[t] == PIPAMD_F_MINUS) pick = -2;  // the kernel pivots on it
    if (pick != -2) {
      // chercher(Critic) then chercher(Unknown), traiter.c:692-693 (flags of the other rows are +/0)
      std::vector<int> cur(nrows);
      for (int t = 0; t < nrows; t++) cur[t] = newflag[t];
      for (int t = 0; t < nrows && pick < 0; t++)
        if (cur[t] == PIPAMD_F_CRITIC) pick = t;
      for (int t = 0; t < nrows && pick < 0; t++)
        if (cur[t] == PIPAMD_F_UNKNOWN) pick = t;
    }
    // drop the sub-jobs' blocks: nothing was allocated in this region after them
    if (f.sub_count) q.top = (size_t)jobs_[f.sub_begin].rows_off - q.region_off;
    f.sub_count = 0;
    jobs_[f.job].status = PIPAMD_ST_RUN;
    f.phase = PH_RUN;
    if (pick < 0) return;  // a negative row to pivot on, or every sign settled
    split(i, pick, newflag);
  }

  // the quast forks on the sign of an undecided row (traiter.c:695-759)
  void split(int i, int t, const std::vector<int> &newflag) {
    Prob &q = P_[i];
    Frame &f = q.stack.back();
    if (f.nparm >= PIPAMD_MAXPARM) fail(i, PIPAMD_ST_INTERNAL);
    const int np = f.nparm, pivi = f.rows[t];
    const std::vector<E> &rv = f.rowvals[t];
    Frame c;
    c.kind = K_NODE;
    c.phase = PH_RUN;
    c.nparm = np;
    c.ni = f.ni;
    c.bigparm = f.bigparm;
    c.flags = f.flags;
    c.mark = q.top;
    c.job = new_job(i, q.nvar, np, f.ni, f.bigparm, f.flags, jobs_[f.job].S, jobs_[f.job].W);
    const PipJob &pj = jobs_[f.job];  // (taken after new_job: the job table may have moved)
    PipJob &cj = jobs_[c.job];
    if (cj.L != pj.L || cj.S != pj.S || cj.W != pj.W) fail(i, PIPAMD_ST_INTERNAL);
    cj.ldet = pj.ldet;
    memcpy(cj.det, pj.det, sizeof pj.det);
    clones_.push_back(pj.rows_off);
    clones_.push_back(cj.rows_off);
    clones_.push_back((i64)pj.L * EW + (i64)pj.L + (i64)pj.S * pj.W * EW);
    tape_push(i, S_IF, (E)0, (E)0);
    tape_push(i, S_FORM, (E)(np + 1), (E)0);
    E g = 0;
    for (int j = 0; j < np; j++) g = gcd(g, rv[1 + j]);
    if (!(f.flags & PIPAMD_T_INT)) g = gcd(g, rv[0]);
    const int nc = f.ctx.nc;
    f.ctx.reserve(nc + 1, np + 2);
    for (int j = 0; j < np; j++) {
      f.ctx.at(nc, j) = cquo(rv[1 + j], g);
      tape_push(i, S_VAL, f.ctx.at(nc, j), (E)1);
    }
    f.ctx.at(nc, np) = (f.flags & PIPAMD_T_INT) ? floordiv(rv[0], g) : cquo(rv[0], g);
    tape_push(i, S_VAL, f.ctx.at(nc, np), (E)1);
    // the clone pass runs before the patch pass, so the copy still has the flags from before this
    // compa_test: give it the refreshed ones (expanser copies them, traiter.c:717), then Plus
    for (size_t k = 0; k < f.rows.size(); k++)
      if ((int)k != t && newflag[k] >= 0) patch_flag(cj, f.rows[k], newflag[k]);
    patch_flag(cj, pivi, PIPAMD_F_PLUS);
    c.ctx = f.ctx;
    c.ctx.nc = nc + 1;
    f.split_row = pivi;
    q.stack.push_back(c);  // (f is invalid from here on)
  }

  // parametric Gomory cut (integrer.c:487-520): context and tape on the host, row appended by patch
  void parm_cut(int i) {
    Prob &q = P_[i];
    Frame &f = q.stack.back();
    PipJob &pj = jobs_[f.job];
    if (pj.tflags & PIPAMD_T_DEEPEST) throw (int)PIPAMD_E_TOOLARGE;
    const E *g = gathered(f.job);
    const int nvar = q.nvar, ni = f.ni;
    int nparm = f.nparm;
    const int ncol = nvar + nparm + 1, nligne = nvar + ni;
    if (ncol >= PIPAMD_MAXCOL) fail(i, PIPAMD_ST_MAXCOL);
    const E D = g[1];
    const E *r = g + 2;
    std::vector<E> cut(ncol + 1);
    bool ok_var = false, ok_parm = false;
    for (int j = 0; j < nvar; j++) {
      cut[j] = fmod_(r[j], D);
      if (cut[j] > 0) ok_var = true;
    }
    cut[nvar] = wneg(fmod_(wneg(r[nvar]), D));
    for (int j = nvar + 1; j < ncol; j++) {
      if (j == f.bigparm) {
        cut[j] = 0;
        continue;
      }
      cut[j] = wneg(fmod_(wneg(r[j]), D));
      if (cut[j] != 0) ok_parm = true;
    }
    cut[ncol] = D;
    if (!ok_parm) throw (int)PIPAMD_E_TOOLARGE;  // only reached with options the Tree handles
    typename Ctx::Quotient k{std::vector<E>(cut.begin() + nvar + 1, cut.begin() + ncol), cut[nvar], D};
    int parm = f.ctx.find_quotient(nparm, k);
    if (parm == -1) {  // a new parameter on this problem's tape and context
      Ctx::announce_quotient([&](int kind, E a, E b) { tape_push(i, kind, a, b); }, nparm, k);
      parm = f.ctx.define_quotient(nparm, k);
      nparm++;
    }
    if (!ok_var) fail(i, PIPAMD_ST_INTERNAL);  // assert(ok_var), integrer.c:499
    const int newcol = nvar + 1 + parm;
    if (ni >= pj.S || nligne >= pj.L || nvar + nparm + 1 > pj.W) throw (int)PIPAMD_E_TOOLARGE;
    std::vector<E> row(pj.W, 0);
    for (int j = 0; j < ncol; j++) row[j] = cut[j];
    row[newcol] = wadd(row[newcol], cut[ncol]);
    patch_vals((size_t)pj.vals_off + (size_t)ni * pj.W * EW, row.data(), row.size());
    patch_vals((size_t)pj.rows_off + (size_t)nligne * EW, &D, 1);
    patch_flag(pj, nligne, PIPAMD_F_MINUS);
    const int slot = ni;
    patch32(((size_t)pj.rows_off + (size_t)pj.L * EW) * 2 + pj.L + nligne, &slot, 1);
    f.ni = ni + 1;
    f.nparm = nparm;
    pj.ni = f.ni;
    pj.nparm = nparm;
    pj.ncut++;
    pj.tflags &= ~PIPAMD_T_STATE;
    pj.status = PIPAMD_ST_RUN;
  }
};
typedef ForestT<i64> Forest;

}  // namespace

// =========================================================================== device tree
// Small problems: the whole traiter() call tree on the device (pip_quast.hip), one wave per problem.
// Fills res[i] for the problems it finishes; the others keep rc == PIPAMD_E_TOOLARGE ("next path").
namespace {
// device buffer `slot` of the engine, at least `bytes` large (kept between calls; grown with headroom, given back
// once eight calls in a row have needed less than a quarter of it while more than 256 MB would stay pinned for nothing:
// a large chunk followed by a small remainder, or large and small calls taking turns, keep their buffer -- every
// hipFree waits for the whole device)
template <class T>
T *dt_buffer(pipamd_engine *e, int slot, size_t bytes) {
  bool oversized = e->dt_cap[slot] > 4 * bytes + ((size_t)256 << 20);
  e->dt_small[slot] = oversized ? e->dt_small[slot] + 1 : 0;
  oversized = oversized && e->dt_small[slot] >= 8;
  if (bytes > e->dt_cap[slot] || oversized) {
    e->dt_small[slot] = 0;
    if (e->dt_buf[slot]) hipFree(e->dt_buf[slot]);
    e->dt_buf[slot] = nullptr;
    e->dt_cap[slot] = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&e->dt_buf[slot], want) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc(&e->dt_buf[slot], bytes) != hipSuccess) {
        e->dt_buf[slot] = nullptr;
        throw (int)PIPAMD_E_HIP;
      }
      e->dt_cap[slot] = bytes;
    } else {
      e->dt_cap[slot] = want;
    }
  }
  return (T *)e->dt_buf[slot];
}

// capacities for one problem; false: not a shape for the device tree
bool quast_caps(const pipamd_problem &p, QCaps &c) {
  const int ncol = p.nvar + p.nparm + 1;
  if (p.nvar < 0 || p.nparm < 0 || p.ni < 0 || p.nc < 0 || (p.ni && !p.ineq) || (p.nc && !p.ctx)) return false;
  if (p.bigparm >= ncol || (p.bigparm >= 0 && p.bigparm <= p.nvar)) return false;
  // at most 64 columns (a lane per column) and 104 inequalities: up to 128 real rows with the cuts (tab_sort_rows has a
  // lane per row up to 64 rows and two rows per lane beyond -- not with Compute_dual, whose `pos` table has 64 entries)
  if (ncol > 64 || p.ni > 104 || p.ni + p.nvar == 0) return false;
  // Room for 10 quotients of parametric cuts, 24 cut rows and 24 nested forks.  (Smaller reserves -- 4 / 8 / 8:
  // 11 KB of LDS instead of 26 KB, twice the problems per CU -- made the launch of 10k problems 27 % shorter,
  // but every problem that then runs out of room costs the host schedulers milliseconds.)
  const int newp = p.nparm ? std::min(10, 64 - ncol) : 0;
  const int depth = p.nparm ? 24 : 0;
  c.W = ncol + newp;
  c.S = p.ni <= 56 ? std::min(64, p.ni + 24) : std::min(128, p.ni + 24);
  c.R = (p.nvar + c.S + 1) & ~1;  // (even: the LDS image and the stack frames are whole 16-byte units in either flavour)
  c.CW = p.nparm + newp + 1;
  c.CR = p.nc + 2 * newp + depth + 2;
  c.SS = std::min(64, c.CR + 1 + 16);
  if (c.CR + 1 > c.SS) return false;
  c.SR = (c.CW - 1 + c.SS + 1) & ~1;
  c.depth = depth;
  c.cells = 4096;  // SOL_SIZE, type.h:33
  return true;
}
void quast_caps_max(QCaps &a, const QCaps &b) {
  a.R = std::max(a.R, b.R);
  a.S = std::max(a.S, b.S);
  a.W = std::max(a.W, b.W);
  a.CR = std::max(a.CR, b.CR);
  a.CW = std::max(a.CW, b.CW);
  a.SR = std::max(a.SR, b.SR);
  a.SS = std::max(a.SS, b.SS);
  a.depth = std::max(a.depth, b.depth);
  a.cells = std::max(a.cells, b.cells);
}

template <class E>
void device_tree_chunk(pipamd_engine *e, const std::vector<int> &idx, const pipamd_problem *probs, const QCaps &cap, int qflags,
                       std::vector<FResultT<E>> &res, int *served, int *handed_back) {
  constexpr int EW = (int)(sizeof(E) / 8), EBITS = 64 * EW;
  const int n = (int)idx.size();
  const bool stats = getenv("PIPAMD_FOREST_STATS") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  std::vector<QProb> qp(n);
  size_t words = 0;
  for (int k = 0; k < n; k++) {
    const pipamd_problem &p = probs[idx[k]];
    qp[k] = QProb{(long long)words, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, qflags, 0};
    words += (size_t)p.ni * (p.nvar + p.nparm + 1) + (size_t)p.nc * (p.nparm + 1);
  }
  if (words * sizeof(i64) > e->dt_host_cap) {  // pinned staging buffer, kept between calls
    if (e->dt_host) hipHostFree(e->dt_host);
    e->dt_host = nullptr;
    e->dt_host_cap = 0;
    const size_t want = words * sizeof(i64) + words * sizeof(i64) / 4 + 4096;
    HIPTHROW(hipHostMalloc(&e->dt_host, want, hipHostMallocDefault));
    e->dt_host_cap = want;
  }
  i64 *in = (i64 *)e->dt_host;
  for (int k = 0; k < n; k++) {
    const pipamd_problem &p = probs[idx[k]];
    const size_t na = (size_t)p.ni * (p.nvar + p.nparm + 1), ncx = (size_t)p.nc * (p.nparm + 1);
    if (na) memcpy(in + qp[k].in_off, p.ineq, na * sizeof(i64));
    if (ncx) memcpy(in + qp[k].in_off + na, p.ctx, ncx * sizeof(i64));
  }
  const size_t frame = pipk_quast_frame_words(&cap, EBITS);  // entries
  QProb *d_prob = dt_buffer<QProb>(e, 0, sizeof(QProb) * n);
  i64 *d_in = dt_buffer<i64>(e, 1, sizeof(i64) * words);
  i64 *d_stack = dt_buffer<i64>(e, 2, sizeof(E) * frame * (size_t)cap.depth * n);
  i64 *d_cells = dt_buffer<i64>(e, 3, sizeof(E) * 3 * (size_t)cap.cells * n);
  int *d_out = dt_buffer<int>(e, 4, sizeof(int) * Q_OUT * n);
  const double t1 = now();
  HIPTHROW(hipMemcpy(d_prob, qp.data(), sizeof(QProb) * n, hipMemcpyHostToDevice));
  HIPTHROW(hipMemcpyAsync(d_in, in, sizeof(i64) * words, hipMemcpyHostToDevice, 0));
  HIPTHROW(pipk_launch_quast(d_prob, d_in, d_stack, d_cells, d_out, n, &cap, EBITS, 0));
  std::vector<int> out(Q_OUT * (size_t)n);
  HIPTHROW(hipMemcpy(out.data(), d_out, sizeof(int) * Q_OUT * n, hipMemcpyDeviceToHost));
  const double t2 = now();
  std::vector<i64> off(n + 1, 0);
  for (int k = 0; k < n; k++) off[k + 1] = off[k] + (out[Q_OUT * k] == Q_DONE ? out[Q_OUT * k + 1] : 0);
  std::vector<E> packed(3 * (size_t)off[n]);
  if (off[n]) {
    i64 *d_off = dt_buffer<i64>(e, 5, sizeof(i64) * (n + 1));
    i64 *d_packed = dt_buffer<i64>(e, 6, sizeof(E) * 3 * (size_t)off[n]);
    HIPTHROW(hipMemcpy(d_off, off.data(), sizeof(i64) * (n + 1), hipMemcpyHostToDevice));
    HIPTHROW(pipk_launch_quast_pack(d_cells, d_off, d_packed, n, cap.cells, EBITS, 0));
    HIPTHROW(hipMemcpy(packed.data(), d_packed, sizeof(E) * packed.size(), hipMemcpyDeviceToHost));
  }
  const double t3 = now();
  if (stats) {
    int why[5] = {0, 0, 0, 0, 0}, tmax = 0;
    long long tsum = 0;
    for (int k = 0; k < n; k++) {
      for (int b = 0; b < 5; b++)
        if (out[Q_OUT * k + 3] >> b & 1) why[b]++;
      if (out[Q_OUT * k] != Q_FALLBACK) tmax = std::max(tmax, out[Q_OUT * k + 4]);
      tsum += out[Q_OUT * k + 4];
      if (out[Q_OUT * k] == Q_FALLBACK)
        fprintf(stderr, "[device tree] problem %d handed back: why %d, %d pivots, %d cells, %.3f ms\n", idx[k], out[Q_OUT * k + 3],
                out[Q_OUT * k + 2], out[Q_OUT * k + 5], out[Q_OUT * k + 4] * 1e-5);
    }
    fprintf(stderr, "[device tree] handed back for: overflow %d, rows %d, tape %d, stack %d, other %d; wave time mean %.3f ms, longest finished %.3f ms\n",
            why[0], why[1], why[2], why[3], why[4], tsum * 1e-5 / n, tmax * 1e-5);
  }
  if (stats)
    fprintf(stderr, "[device tree] %d problems, LDS %zu B, frame %zu words x %d; pack+alloc %.2f ms, copy+kernel %.2f ms, tape %.2f ms (%lld cells)\n",
            n, pipk_quast_lds_bytes(&cap, EBITS), frame, cap.depth, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (long long)off[n]);
  for (int k = 0; k < n; k++) {
    FResultT<E> &r = res[idx[k]];
    const int st = out[Q_OUT * k];
    if (st == Q_FALLBACK) {
      ++*handed_back;
      continue;
    }
    ++*served;
    r.rc = PIPAMD_OK;
    r.status = 0;
    r.pivots = out[Q_OUT * k + 2];
    r.is_void = st == Q_VOID;
    r.tape.clear();
    if (st == Q_DONE) {
      r.tape.resize((size_t)(off[k + 1] - off[k]));
      const E *c = packed.data() + 3 * (size_t)off[k];
      for (size_t t = 0; t < r.tape.size(); t++) r.tape[t] = CellT<E>{(int)c[3 * t], c[3 * t + 1], c[3 * t + 2]};
    }
  }
}

template <class E>
void device_tree(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut, std::vector<FResultT<E>> &res,
                 int *served, int *handed_back, int qflags = 0) {
  constexpr int EBITS = 64 * (int)(sizeof(E) / 8);
  // LDS a problem's image may take: 96 KB in 64 bits (the tall shapes of up to 128 real rows; the usual ones are a quarter
  // of that); the 128-bit flavour, whose image is twice the size, may have a CU's worth
  const size_t lds_limit = EBITS == 128 ? (size_t)150 * 1024 : (size_t)96 * 1024;
  struct Hold {  // the engine's device-tree buffers serve one call at a time
    pthread_mutex_t *m;
    explicit Hold(pthread_mutex_t *mm) : m(mm) { pthread_mutex_lock(m); }
    ~Hold() { pthread_mutex_unlock(m); }
  } hold(&e->dt_lock);
  // chunks of problems whose stack + tape regions fit the budget (PIPAMD_FOREST_ARENA_MB, default 4096)
  size_t budget = (size_t)4096 << 20;
  if (const char *mb = getenv("PIPAMD_FOREST_ARENA_MB")) budget = (size_t)strtoull(mb, nullptr, 10) << 20;
  std::vector<int> idx;
  QCaps cap;
  memset(&cap, 0, sizeof cap);
  auto flush = [&]() {
    if (idx.empty()) return;
    cap.deepest = deepest_cut ? 1 : 0;
    cap.simplify = simplify ? 1 : 0;  // the kernel simplifies the rows as it loads them
    try {
      device_tree_chunk<E>(e, idx, probs, cap, qflags, res, served, handed_back);
    } catch (int) {  // allocation or launch failure: the chunk's problems go to the next path
      (void)hipGetLastError();
    }
    idx.clear();
    memset(&cap, 0, sizeof cap);
  };
  // Problems of similar size share a launch: every problem of a launch gets the LDS image of the largest, so a
  // batch of tiny problems with one large one among them would run at the large one's occupancy.  Size classes
  // of 8 KB of LDS, smallest first.
  std::vector<std::pair<int, int>> order;  // (size class, problem)
  order.reserve(n);
  for (int i = 0; i < n; i++) {
    QCaps c;
    memset(&c, 0, sizeof c);
    if (!quast_caps(probs[i], c)) continue;
    const size_t lds = pipk_quast_lds_bytes(&c, EBITS);
    if (lds > lds_limit) continue;
    order.emplace_back((int)(lds / 8192), i);
  }
  std::stable_sort(order.begin(), order.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
  int cls = -1;
  for (const auto &oc : order) {
    const int i = oc.second;
    QCaps c;
    memset(&c, 0, sizeof c);
    quast_caps(probs[i], c);
    if (oc.first != cls) {
      flush();
      cls = oc.first;
    }
    QCaps m = cap;
    quast_caps_max(m, c);
    if (pipk_quast_lds_bytes(&m, EBITS) > lds_limit) {  // (the maxima of several shapes of one class together)
      flush();
      m = c;
    }
    const size_t per = sizeof(E) * (pipk_quast_frame_words(&m, EBITS) * (size_t)m.depth + 3 * (size_t)m.cells);
    if (!idx.empty() && per * (idx.size() + 1) > budget) {
      flush();
      m = c;
    }
    cap = m;
    idx.push_back(i);
  }
  flush();
}
}  // namespace

template <class E>
static bool device_tree_one(pipamd_engine *e, const pipamd_problem &p, int simplify, int deepest_cut, int qflags,
                            std::vector<CellT<E>> &tape, bool *is_void, int64_t *pivots) {
  if (e->no_device_tree || getenv("PIPAMD_NO_DEVICE_TREE")) return false;
  std::vector<FResultT<E>> res(1);
  res[0].rc = PIPAMD_E_TOOLARGE;
  int served = 0, back = 0;
  device_tree<E>(e, 1, &p, simplify, deepest_cut, res, &served, &back, qflags);
  pthread_mutex_lock(&e->dt_lock);
  e->dt_served = served;  // (pipamd_last_device_tree also answers for the one-problem entries)
  e->dt_fallback = back;
  pthread_mutex_unlock(&e->dt_lock);
  if (res[0].rc != PIPAMD_OK) return false;
  tape.swap(res[0].tape);
  *is_void = res[0].is_void;
  *pivots = res[0].pivots;
  return true;
}

template <class E, class CELL>
static void device_tree_many(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut,
                             std::vector<char> &served, CELL **cells, size_t *n_cells, int *rcs, int *statuses,
                             int64_t *pivots) {
  e->dt_served = e->dt_fallback = 0;
  if (e->no_device_tree || getenv("PIPAMD_NO_DEVICE_TREE") || n <= 0) return;
  std::vector<FResultT<E>> res(n);
  for (auto &r : res) r.rc = PIPAMD_E_TOOLARGE;
  device_tree<E>(e, n, probs, simplify, deepest_cut, res, &e->dt_served, &e->dt_fallback);
  for (int i = 0; i < n; i++) {
    if (res[i].rc != PIPAMD_OK) continue;
    served[i] = 1;
    cells[i] = nullptr;
    n_cells[i] = 0;
    if (statuses) statuses[i] = 0;
    if (pivots) pivots[i] = res[i].pivots;
    rcs[i] = res[i].is_void ? PIPAMD_OK : export_tape(res[i].tape, &cells[i], &n_cells[i]);
  }
}

// Many problems: the device tree first (small problems), then the lock-step Forest; the few that need
// a rare path are finished by the Tree.
template <class E>
static void lockstep_device_tree(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut,
                                 std::vector<FResultT<E>> &res) {
  if (!e->no_device_tree && !getenv("PIPAMD_NO_DEVICE_TREE"))
    device_tree<E>(e, n, probs, simplify, deepest_cut, res, &e->dt_served, &e->dt_fallback);
}

template <class E, class CELL>
static int lockstep_any(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify, int deepest_cut, CELL **cells,
                        size_t *n_cells, int *rcs, int *statuses, int64_t *pivots) {
  typedef FResultT<E> FResult;
  typedef ForestT<E> Forest;
  typedef TreeT<E> Tree;
  if (!e || n < 0 || (n && (!probs || !cells || !n_cells || !rcs))) return PIPAMD_E_INVALID;
  if (hipSetDevice(e->device) != hipSuccess) return PIPAMD_E_HIP;
  std::vector<FResult> res(n);
  for (auto &r : res) r.rc = PIPAMD_E_TOOLARGE;  // until a path has served it
  e->dt_served = e->dt_fallback = 0;
  lockstep_device_tree<E>(e, n, probs, simplify, deepest_cut, res);
  std::vector<int> rest;
  for (int i = 0; i < n; i++)
    if (res[i].rc == PIPAMD_E_TOOLARGE) rest.push_back(i);
  if (!deepest_cut && !rest.empty()) {
    std::vector<pipamd_problem> rp(rest.size());
    for (size_t k = 0; k < rest.size(); k++) rp[k] = probs[rest[k]];
    const int nr = (int)rest.size();
    // The forest reserves a worst-case region per problem; batches whose regions add up to more
    // than the arena budget go through it in chunks (PIPAMD_FOREST_ARENA_MB, default 4096).
    size_t budget = (size_t)4096 << 20;
    if (const char *mb = getenv("PIPAMD_FOREST_ARENA_MB")) budget = (size_t)strtoull(mb, nullptr, 10) << 20;
    try {
      Forest f(e->device);
      for (int lo = 0; lo < nr;) {
        size_t bytes = 0;
        int hi = lo;
        while (hi < nr && (hi == lo || bytes + Forest::region_words(rp[hi]) * sizeof(i64) <= budget))
          bytes += Forest::region_words(rp[hi++]) * sizeof(i64);
        std::vector<FResult> part;
        try {
          f.solve(hi - lo, rp.data() + lo, simplify, part);
          for (int i = lo; i < hi; i++) res[rest[i]] = std::move(part[i - lo]);
        } catch (int code) {
          // an allocation or launch failure of this chunk: its problems go to the per-problem tree
          (void)hipGetLastError();
        }
        lo = hi;
      }
    } catch (int code) {
      if (code == PIPAMD_E_HIP) return code;  // not even a stream
    }
  }
  Tree *fallback = nullptr;
  int rc_all = PIPAMD_OK;
  for (int i = 0; i < n; i++) {
    cells[i] = nullptr;
    n_cells[i] = 0;
    if (res[i].rc == PIPAMD_E_TOOLARGE) {
      int st = 0;
      int64_t pv = 0;
      const pipamd_problem &p = probs[i];
      try {
        if (!fallback) fallback = new Tree(e, deepest_cut);
        rcs[i] = solve_one(*fallback, p.nvar, p.nparm, p.ni, p.nc, p.bigparm, p.nq, p.ineq, p.ctx, simplify, deepest_cut,
                           &cells[i], &n_cells[i], &st, &pv);
      } catch (int code) {
        rcs[i] = code;
      }
      if (statuses) statuses[i] = st;
      if (pivots) pivots[i] = pv;
      continue;
    }
    rcs[i] = res[i].rc;
    if (statuses) statuses[i] = res[i].status;
    if (pivots) pivots[i] = res[i].pivots;
    if (res[i].rc || res[i].is_void) continue;
    rcs[i] = export_tape(res[i].tape, &cells[i], &n_cells[i]);
  }
  delete fallback;
  return rc_all;
}

extern "C" int pipamd_solve_tableaux_lockstep(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify,
                                              int deepest_cut, pipamd_sol_cell **cells, size_t *n_cells, int *rcs,
                                              int *statuses, int64_t *pivots) {
  return lockstep_any<i64, pipamd_sol_cell>(e, n, probs, simplify, deepest_cut, cells, n_cells, rcs, statuses, pivots);
}
// The same on 128-bit entries (the overflow-safe flavour, include/piplib/piplib.h:42-88): ForestT<__int128> -- device tableaux,
// contexts, cut rows and tape cells 128-bit --, the rare paths finished by a TreeT<__int128>.
extern "C" int pipamd_solve_tableaux_lockstep128(pipamd_engine *e, int n, const pipamd_problem *probs, int simplify,
                                                 int deepest_cut, pipamd_sol_cell128 **cells, size_t *n_cells, int *rcs,
                                                 int *statuses, int64_t *pivots) {
  return lockstep_any<i128, pipamd_sol_cell128>(e, n, probs, simplify, deepest_cut, cells, n_cells, rcs, statuses, pivots);
}
