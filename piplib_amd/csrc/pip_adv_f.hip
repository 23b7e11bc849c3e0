// piplib_amd/csrc/pip_adv_f.hip -- group F of the pivot kernel's instantiations (pip_adv_inst.h)
#include "pip_advance.h"
PIP_ADV_GROUP_F(PIP_ADV_DEFINE)
