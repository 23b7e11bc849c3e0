// piplib_amd/csrc/pip_adv_b.hip -- group B of the pivot kernel's instantiations (pip_adv_inst.h)
#include "pip_advance.h"
PIP_ADV_GROUP_B(PIP_ADV_DEFINE)
