// piplib_amd/csrc/pip_adv_inst.h -- the instantiations of pip_advance_kernel<T, NCH, NW, GM, SC, FULL>, in four
// groups of similar compile time (one translation unit each: pip_adv_a.hip ... pip_adv_d.hip).  Keep in step with
// launch_advance_w / launch_static in pip_kernels.hip: an instantiation used there and missing here fails at link time.
#ifndef PIP_ADV_INST_H
#define PIP_ADV_INST_H
// A: one wave per tableau, <= 128 int64 columns, compile-time row capacity, FULL (127 unknowns + constant, no parameters)
#define PIP_ADV_GROUP_A(X) \
  X(i64, 1, 1, false, 64, true) X(i64, 1, 1, false, 96, true) X(i64, 1, 1, false, 112, true) \
  X(i64, 1, 1, false, 128, true) X(i64, 1, 1, false, 160, true)
// B: the same without FULL, and the run-time row capacity
#define PIP_ADV_GROUP_B(X) \
  X(i64, 1, 1, false, 64, false) X(i64, 1, 1, false, 96, false) X(i64, 1, 1, false, 112, false) \
  X(i64, 1, 1, false, 128, false) X(i64, 1, 1, false, 160, false) X(i64, 1, 1, false, 0, false)
// C: four / eight waves per tableau, row tables in HBM, 256 and 512 columns
#define PIP_ADV_GROUP_C(X) \
  X(i64, 1, 4, false, 0, false) X(i64, 1, 8, false, 0, false) X(i64, 1, 4, true, 0, false) \
  X(i64, 2, 1, false, 0, false) X(i64, 2, 4, false, 0, false) X(i64, 2, 4, true, 0, false) \
  X(i64, 4, 1, false, 0, false) X(i64, 4, 4, false, 0, false) X(i64, 4, 4, true, 0, false)
// D: 128-bit entries
#define PIP_ADV_GROUP_D(X) \
  X(i128, 1, 1, false, 0, false) X(i128, 1, 4, false, 0, false) X(i128, 2, 1, false, 0, false) \
  X(i128, 2, 4, false, 0, false) X(i128, 4, 1, false, 0, false) X(i128, 4, 4, false, 0, false) \
  X(i128, 8, 1, false, 0, false) X(i128, 8, 4, false, 0, false)
// F: 128-bit entries, sixteen waves per tableau (a whole CU): the tail launches over a few long tableaux with hundreds of rows
#define PIP_ADV_GROUP_F(X) X(i128, 1, 16, false, 0, false) X(i128, 2, 16, false, 0, false) X(i128, 4, 16, false, 0, false)
#define PIP_ADV_DEFINE(...) template hipError_t launch_advance_t<__VA_ARGS__>(const AdvanceLaunch &);
#endif
