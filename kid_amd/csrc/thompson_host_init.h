// thompson_host_init.h -- host half of thompson_init (M:374-670).
#pragma once
#include "thompson_params.h"

namespace kidmp {
void host_init(int iiwarm, int l_sediment, double set_Nc, Consts &c, Bins &b);
}
