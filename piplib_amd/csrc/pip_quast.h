// piplib_amd/csrc/pip_quast.h -- the device-resident traiter() of pip_quast.hip: launch interface.
#ifndef PIP_QUAST_H
#define PIP_QUAST_H
#include <hip/hip_runtime.h>
#include <stddef.h>

// one problem: its rows sit at input[in_off]: ni x (nvar+nparm+1) inequalities, then nc x (nparm+1) context rows
struct QProb {
  long long in_off;
  int nvar, nparm, ni, nc, bigparm, nq;
  int flags, pad;  // Q_NO_CONTEXT_TEST: traiter() proper (the front ends test the context in a call of its own)
};
enum { Q_NO_CONTEXT_TEST = 1,
       Q_DUAL = 2 };  // Compute_dual (traiter.c:273-294): the list of dual values behind every (rational) solution
// capacities of one launch (every problem of the launch gets the same LDS image and HBM regions)
struct QCaps {
  int R, S, W;    // main tableau: logical rows, real-row slots, columns (W <= 64, S <= 128)
  int CR, CW;     // context: rows, columns (parameters | constant)
  int SR, SS;     // compa_test sub-problems: logical rows, real-row slots (their width is CW)
  int depth;      // frames of the fork stack
  int cells;      // tape cells
  int deepest;    // deepest-cut option (integrer.c:417-438)
  int simplify;   // tab_simplify on the inputs of integer problems (maind.c:190-196)
};
// out[Q_OUT*i]: result; +1: cells; +2: pivots; +3: why a problem was handed back (Q_WHY_* bits);
// +4: run time of the problem's wave in 10 ns units; +5: cells written when it stopped
enum { Q_DONE = 0, Q_VOID = 1, Q_FALLBACK = 2 };
enum { Q_OUT = 6 };
enum {
  Q_WHY_OVERFLOW = 1,  // a 64-bit product / sum overflowed, or the determinant did ("Integer overflow")
  Q_WHY_ROWS = 2,      // rows, columns or context rows reserved for the problem ran out
  Q_WHY_TAPE = 4,      // SOL_SIZE cells
  Q_WHY_STACK = 8,     // fork depth
  Q_WHY_OTHER = 16     // too many parameters, iteration guard, assert(ok_var), ...
};

extern "C" {
// ebits: the entry width of the launch, 64 (the reference's long long build) or 128 (the overflow-safe flavour)
size_t pipk_quast_lds_bytes(const QCaps *c, int ebits);
size_t pipk_quast_frame_words(const QCaps *c, int ebits);  // entries of one stack frame
hipError_t pipk_launch_quast(const QProb *probs, const long long *input, void *stack, void *cells, int *out, int nprob,
                             const QCaps *cap, int ebits, hipStream_t stream);
hipError_t pipk_launch_quast_pack(const long long *cells, const long long *off, long long *packed, int nprob,
                                  int cells_cap, int ebits, hipStream_t stream);
}
#endif
