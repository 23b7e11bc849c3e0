// piplib_amd/csrc/pip_adv_e.hip -- group E: the lean bulk kernel (pip_lean.h), one instantiation per row-capacity class
#include "pip_lean.h"
PIP_LEAN_CLASSES(PIP_LEAN_DEFINE)
