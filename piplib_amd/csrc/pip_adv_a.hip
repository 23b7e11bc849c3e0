// piplib_amd/csrc/pip_adv_a.hip -- group A of the pivot kernel's instantiations (pip_adv_inst.h)
#include "pip_advance.h"
PIP_ADV_GROUP_A(PIP_ADV_DEFINE)
