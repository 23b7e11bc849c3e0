// piplib_amd/csrc/pip_adv_c.hip -- group C of the pivot kernel's instantiations (pip_adv_inst.h)
#include "pip_advance.h"
PIP_ADV_GROUP_C(PIP_ADV_DEFINE)
