// thompson_column.h -- launch interface of the gfx950 column-step kernel
// (mp_thompson, M:1156-3688).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "thompson_params.h"

namespace kidmp {

constexpr int KIDMP_NRATES_ = 36;    // save_dg rates per level, order of M:2967-3119
constexpr int MAX_CONST_SLOTS = 32;  // contexts alive at once per process (constant-memory slots)

// All pointers are device pointers; profiles are x[col*nz + k] (k fastest).  R = the reference's REAL in the
// arithmetic variant that is launched: double (p64) or float (p32n, f32).
template <class R>
struct StepArgsT {
    R *qv, *qc, *qi, *qr, *qs, *qg, *ni, *nr, *nc, *nwfa, *nifa, *t;   // INOUT, M:1168-1170
    const R *p, *dz;                                                    // IN (w1d is inert, M:2797)
    R *ppt;               // [ncol][4] rain, snow, graupel, ice (accumulated, M:1172)
    double *rates;        // nullptr or [ncol][36][nz] (binary64 in every variant: the rates are DOUBLE PRECISION)
    int32_t *nstep;       // nullptr or [ncol][4] rain, ice, snow, graupel
    int32_t cslot;        // slot of this context's Consts in constant memory (upload_consts)
    int32_t iiwarm;       // the context's iiwarm switch: selects the warm-rain instantiation of the kernel
    int32_t aero;         // the context's is_aerosol_aware switch (M:28): selects the aerosol-aware instantiation
    const R *w;           // IN: updraft, read only by activ_ncloud (aerosol-aware contexts, M:2797); may be null otherwise
    Tables tables;
    int64_t ncol;
    int32_t nz;
    R dt;
    int32_t debug_stop;   // 0 = run everything; n = leave after pass n-1 (libkidmp_prof.so, env KIDMP_DEBUG_STOP)
};
typedef StepArgsT<double> StepArgs;

// thompson_column.hip is compiled once per arithmetic variant, each into its own namespace:
//   p64  (real = double, dreal = double), p32n (float state, double rates: the reference as shipped), f32 (all float)
#define KIDMP_DECLARE_VARIANT(ns, R)                                                        \
    namespace ns {                                                                          \
    hipError_t launch_column_step(const StepArgsT<R> &a, hipStream_t s);                    \
    hipError_t upload_consts(int slot, const Consts &c);                                    \
    std::string column_kernel_fingerprint(bool iiwarm);                                     \
    }
KIDMP_DECLARE_VARIANT(p64, double)
KIDMP_DECLARE_VARIANT(p32n, float)
KIDMP_DECLARE_VARIANT(f32, float)
#undef KIDMP_DECLARE_VARIANT

bool generated_consts_match(const Consts &c);   // thompson_consts_gen.h vs the run-time host init
const char *column_kernel_name();

}  // namespace kidmp
