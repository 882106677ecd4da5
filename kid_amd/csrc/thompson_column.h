// thompson_column.h -- launch interface of the gfx950 column-step kernel
// (mp_thompson, M:1156-3688).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "thompson_params.h"

namespace kidmp {

constexpr int KIDMP_NRATES_ = 36;
constexpr int MAX_CONST_SLOTS = 8;   // contexts alive at once per process (constant-memory slots)    // save_dg rates per level, order of M:2967-3119

// All pointers are device pointers; profiles are x[col*nz + k] (k fastest).
struct StepArgs {
    double *qv, *qc, *qi, *qr, *qs, *qg, *ni, *nr, *nc, *nwfa, *nifa, *t;   // INOUT, M:1168-1170
    const double *p, *dz;                                                    // IN (w1d is inert, M:2797)
    double *ppt;          // [ncol][4] rain, snow, graupel, ice (accumulated, M:1172)
    double *rates;        // nullptr or [ncol][36][nz]
    int32_t *nstep;       // nullptr or [ncol][4] rain, ice, snow, graupel
    double *scratch;      // [ncol][nz] work profile owned by the context (block-K rain mvd, pass 1 -> pass 3)
    int32_t cslot;        // slot of this context's Consts in constant memory (upload_consts)
    int32_t iiwarm;       // the context's iiwarm switch: selects the warm-rain instantiation of the kernel
    Tables tables;
    int64_t ncol;
    int32_t nz;
    double dt;
    int32_t debug_stop;   // 0 = run everything; n = leave after pass n-1 (profiling aid, env KIDMP_DEBUG_STOP)
};

hipError_t launch_column_step(const StepArgs &a, hipStream_t s);
hipError_t upload_consts(int slot, const Consts &c);
bool generated_consts_match(const Consts &c);   // thompson_consts_gen.h vs the run-time host init
const char *column_kernel_name();
std::string column_kernel_fingerprint(bool iiwarm);   // source hash + register/LDS/scratch use of the nz <= 120 kernel

}  // namespace kidmp
