/* oracle/batchfmt.h -- TEST INFRASTRUCTURE.
 * Binary batch exchange format shared by the reference driver (ref_driver.c),
 * the CPU restatement's CLI (oracle_cli.c) and the Python test helpers
 * (tests/pipbatch.py).  All fields little-endian, naturally packed.
 *
 * input : batch_hdr, then per problem: batch_prob, ni*(nvar+nparm+1) int64
 *         (inequality rows, PIP column order: unknowns | constant | parameters),
 *         nc*(nparm+1) int64 (context rows: parameters | constant).
 * output: batch_out_hdr, then per problem: batch_res, text_len bytes of
 *         sol_edit-format text (sol.c:291-422).
 */
#ifndef ORACLE_BATCHFMT_H
#define ORACLE_BATCHFMT_H
#include <stdint.h>

#define BATCH_MAGIC 0x50495042u /* "BPIP" */

#define BATCH_F_NOTEXT 1u     /* do not emit solution text (timing runs) */
#define BATCH_F_NOSIMPLIFY 2u /* skip tab_simplify even when nq != 0 */
#define BATCH_F_DEEPEST 4u    /* deepest-cut option (piplib.c:53) */

#define BATCH_ST_OK 0
#define BATCH_ST_VOID 1  /* empty context: front end prints "void" */
#define BATCH_ST_ABORT 2 /* the solver called exit(abort_code) */

struct batch_hdr {
  uint32_t magic, count, flags, reserved;
};
struct batch_prob {
  int32_t nvar, nparm, ni, nc, bigparm, nq;
};
struct batch_out_hdr {
  uint32_t magic, count;
  double solve_seconds;
  int64_t total_pivots;
};
struct batch_res {
  int32_t status, abort_code;
  int64_t pivots;
  uint32_t text_len, reserved;
};
#endif
