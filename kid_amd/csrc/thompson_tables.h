// thompson_tables.h -- GPU construction of the Thompson lookup tables
// (table half of thompson_init, M:676-791).
#pragma once
#include <hip/hip_runtime.h>

#include "thompson_params.h"

namespace kidmp {
hipError_t alloc_tables(Tables &t);
void free_tables(Tables &t);
// d_consts/d_bins are device copies; runs the builders on `s` and waits.
hipError_t repack_records(Tables &t, hipStream_t s);
hipError_t build_tables(const Consts *d_consts, const Bins *d_bins, int iiwarm, Tables &t, hipStream_t s);
}
