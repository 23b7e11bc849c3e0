/* tests/refbind/layout_check.c -- TEST INFRASTRUCTURE.
 *
 * Compile-time check of include/piplib_amd.h against the reference's own headers: the constants
 * the C ABI shares with PipLib must have PipLib's values.  Built by oracle/Makefile (target `ref`,
 * only where /root/reference is present) with -I/root/reference/source; a successful compile IS
 * the test, running the binary prints the values for the log.
 *
 * The structures of pip_solve()'s interface (PipMatrix, PipQuast, ...) do not appear in the C ABI
 * any more: pip_solve stays the reference's own code (source/piplib.c) and calls the engine through
 * traiter()'s signature -- see bindings/piplib_traiter_hook.c and INTEGRATION.md.
 */
#include <stdio.h>

#include "pip.h" /* source/pip.h: type.h, sol.h, tab.h, funcall.h */
#include "piplib_amd.h"

_Static_assert(PIPAMD_T_INT == TRAITER_INT, "funcall.h:34");
_Static_assert(PIPAMD_T_DUAL == TRAITER_DUAL, "funcall.h:35");
_Static_assert(PIPAMD_F_UNIT == Unit && PIPAMD_F_PLUS == Plus && PIPAMD_F_MINUS == Minus, "tab.h:55-57");
_Static_assert(PIPAMD_F_ZERO == Zero && PIPAMD_F_CRITIC == Critic && PIPAMD_F_UNKNOWN == Unknown, "tab.h:58-60");
_Static_assert(sizeof(piplib_int_t_dp) == sizeof(int64_t), "the dp flavour's Entier is 64 bits");
_Static_assert(sizeof(((pipamd_sol_cell *)0)->param1) == sizeof(piplib_int_t_dp), "sol.c:52-59 param1/param2");

int main(void) {
  printf("layout_check ok: TRAITER_INT=%d TRAITER_DUAL=%d Unit..Unknown=%d,%d,%d,%d,%d,%d MAXCOL=%d MAXPARM=%d "
         "MAX_DETERMINANT=%d\n",
         TRAITER_INT, TRAITER_DUAL, Unit, Plus, Minus, Zero, Critic, Unknown, MAXCOL, MAXPARM, MAX_DETERMINANT);
  return 0;
}
