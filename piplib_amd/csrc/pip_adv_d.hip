// piplib_amd/csrc/pip_adv_d.hip -- group D of the pivot kernel's instantiations (pip_adv_inst.h)
#include "pip_advance.h"
PIP_ADV_GROUP_D(PIP_ADV_DEFINE)
