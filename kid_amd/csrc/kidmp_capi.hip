// kidmp_capi.hip -- the C ABI of include/kidmp.h over the gfx950 kernels.
// Host-side mirror of the reference's thompson_init / mp_thompson pair
// (M:374, M:1156) plus the batched form of the KiD adapter loop (W:54-246).
#include <hip/hip_runtime.h>

#include <rccl/rccl.h>              // types only: the library is dlopen'ed by kidmp_init_multi (a one-GPU host needs no RCCL)
#include <dlfcn.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kidmp.h"
#include "thompson_column.h"
#include "thompson_host_init.h"
#include "thompson_tables.h"
#include "table_cache.h"
#include "fastmath.h"

using namespace kidmp;

constexpr int HOST_NBUF = 3;

struct kidmp_ctx {
    kidmp_cfg cfg{};
    Consts hc{};
    Bins hb{};
    Consts *d_consts = nullptr;
    Bins *d_bins = nullptr;
    Tables tables{};
    bool ready = false;
    double init_s = 0.;
    std::string err;
    // staging for the host-array entries: a ring of HOST_NBUF column chunks in HBM, one stream per direction and
    // one for the kernel, so that the upload of chunk i+1, the step of chunk i and the download of chunk i-1 overlap
    double *d_stage = nullptr;
    size_t stage_bytes = 0;
    hipStream_t stream = nullptr;                    // the context's compute stream
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_up[HOST_NBUF] = {}, ev_step[HOST_NBUF] = {}, ev_down[HOST_NBUF] = {};
    int64_t host_chunk = 0;                          // columns per chunk; 0 = chosen per call (kidmp_set_host_chunk)
    int debug_stop = 0;
    int cslot = -1;
    // partial sums of kidmp_reduce_rates_device, accumulators of kidmp_sanity_device
    double *d_red = nullptr;
    size_t red_elems = 0;
    unsigned long long *d_sanity = nullptr;
    // exact (fixed-point) domain sums of the surface precipitation: KIDMP_PPT_LIMBS 64-bit accumulators
    unsigned long long *d_acc = nullptr;
    std::string fingerprint;
};

namespace {

thread_local std::string g_err;
std::mutex g_slot_mu;
bool g_slot_used[MAX_CONST_SLOTS] = {};

int take_slot()
{
    std::lock_guard<std::mutex> g(g_slot_mu);
    for (int i = 0; i < MAX_CONST_SLOTS; ++i)
        if (!g_slot_used[i]) { g_slot_used[i] = true; return i; }
    return -1;
}
void give_slot(int i)
{
    std::lock_guard<std::mutex> g(g_slot_mu);
    if (i >= 0 && i < MAX_CONST_SLOTS) g_slot_used[i] = false;
}

int fail(kidmp_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    g_err = msg;
    return code;
}
int hipfail(kidmp_ctx *c, hipError_t e, const char *what)
{
    return fail(c, KIDMP_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPTRY(c, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hipfail((c), e_, #x); } while (0)

// Every entry point that touches the device runs with the context's device current and puts the caller's
// device back on exit: the caller (torch, a Fortran host driving several GPUs) may have another one selected,
// and hipMalloc / kernel launches / the __constant__ slot all bind to the current device.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define GUARD(c) DeviceGuard guard_((c)->cfg.device); if (guard_.err != hipSuccess) return hipfail((c), guard_.err, "hipSetDevice")

// A device pointer handed to a device entry must live on the context's GPU: a buffer of another GPU would be
// reached through peer access at best and fault at worst.
int check_on_device(kidmp_ctx *c, const void *p, const char *what)
{
    if (!p) return KIDMP_OK;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return KIDMP_OK; }   // unregistered: let the launch decide
    if (at.type == hipMemoryTypeDevice && at.device != c->cfg.device)
        return fail(c, KIDMP_EINVAL, std::string("kidmp: ") + what + " lives on device " + std::to_string(at.device)
                                     + ", the context is bound to device " + std::to_string(c->cfg.device));
    return KIDMP_OK;
}

// the non-aerosol defaults of M:958-964 in the state's own arithmetic (REAL expressions of the reference)
template <class T>
__global__ void k_default_aerosols(int64_t n, T Nt_c, const T *__restrict__ qv, const T *__restrict__ t,
                                   const T *__restrict__ p, T *__restrict__ nc, T *__restrict__ nwfa, T *__restrict__ nifa)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T rho = T(0.622) * p[i] / (T(Rgas) * t[i] * (qv[i] + T(0.622)));   // M:959
    nc[i] = Nt_c / rho;                                                      // M:960
    nwfa[i] = T(11.1E6) / rho;                                               // M:961
    nifa[i] = T(naIN1) * T(0.01) / rho;                                      // M:962
}

// out4[s] = sum over columns of ppt[col][s]; one block, fixed order => reproducible
__global__ void k_reduce_ppt(int64_t ncol, const double *__restrict__ ppt, double *__restrict__ out4)
{
    __shared__ double sh[256][4];
    double acc[4] = {0., 0., 0., 0.};
    for (int64_t c = threadIdx.x; c < ncol; c += blockDim.x)
        for (int s = 0; s < 4; ++s) acc[s] += ppt[c * 4 + s];
    for (int s = 0; s < 4; ++s) sh[threadIdx.x][s] = acc[s];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (int(threadIdx.x) < w)
            for (int s = 0; s < 4; ++s) sh[threadIdx.x][s] += sh[threadIdx.x + w][s];
        __syncthreads();
    }
    if (threadIdx.x < 4) out4[threadIdx.x] = sh[0][threadIdx.x];
}

// Exact domain sums of ppt[col][0..3] (the nx-means of W:248-303 are these / nx), independent of the order of the
// additions and therefore of how the columns are sharded over devices or chunks: every value is cut into 32-bit
// pieces on a fixed-point grid (least significant bit 2**-128, six 64-bit limbs per species, limb j weighs
// 2**(32 j - 128)) and the pieces are added with integer atomics.  Integer addition is associative, so one GPU, eight
// GPUs or two contexts on one GPU end with the same 24 limbs, bit for bit; an all-reduce(SUM) of int64 limbs over the
// devices keeps that.  Range: |x| < 2**32; bits below 2**-128 (3e-39) are dropped; non-finite values are ignored.
constexpr int ACC_LIMBS = 6, ACC_N = 4 * ACC_LIMBS;
static_assert(ACC_N == KIDMP_PPT_LIMBS, "include/kidmp.h");
template <class T>
__global__ void k_ppt_exact(int64_t ncol, const T *__restrict__ ppt, unsigned long long *__restrict__ acc)
{
    __shared__ unsigned long long sh[ACC_N];
    if (threadIdx.x < ACC_N) sh[threadIdx.x] = 0ull;
    __syncthreads();
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < 4 * ncol; i += int64_t(gridDim.x) * blockDim.x) {
        const double x = double(ppt[i]);
        const int sp = int(i & 3);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
        const int e = int((bits >> 52) & 0x7ff);
        if (e == 0 || e >= 1023 + 32) continue;                       // zero / subnormal: below the grid; >= 2**32, inf, nan: ignored
        unsigned long long m = (bits & ((1ull << 52) - 1)) | (1ull << 52);   // x = m * 2**(e - 1075)
        int shft = e - 1075 + 128;                                    // position of m's bit 0 on the grid
        if (shft < 0) {
            if (shft <= -53) continue;
            m >>= -shft;
            shft = 0;
        }
        const int j = shft >> 5, r = shft & 31;                       // shft <= 107: j <= 3, pieces land in limbs j .. j+2 <= 5
        const unsigned long long lo = m << r, hi = r ? (m >> (64 - r)) : 0ull;
        unsigned long long pc[3] = {lo & 0xffffffffull, lo >> 32, hi};
        const bool neg = (bits >> 63) != 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (pc[q]) atomicAdd(&sh[sp * ACC_LIMBS + j + q], neg ? (0ull - pc[q]) : pc[q]);   // two's complement
    }
    __syncthreads();
    if (threadIdx.x < ACC_N && sh[threadIdx.x]) atomicAdd(&acc[threadIdx.x], sh[threadIdx.x]);
}
template <class T>
hipError_t launch_ppt_exact(int64_t ncol, const T *ppt, unsigned long long *acc, hipStream_t s)
{
    if (ncol <= 0) return hipSuccess;
    int64_t g = (4 * ncol + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(k_ppt_exact<T>, dim3((unsigned)g), dim3(256), 0, s, ncol, ppt, acc);
    return hipGetLastError();
}

// Domain sums of the rate diagnostics: part[chunk][r*nz+k] = sum over the chunk's columns (fixed order), then
// out[r*nz+k] = sum over chunks (fixed order) => bitwise reproducible for a given ncol.
constexpr int RED_CHUNKS = 128;
__global__ void k_reduce_rates_part(int64_t ncol, int n, const double *__restrict__ rates, double *__restrict__ part)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // r*nz + k
    if (i >= n) return;
    const int64_t per = (ncol + RED_CHUNKS - 1) / RED_CHUNKS;
    const int64_t c0 = int64_t(blockIdx.y) * per, c1 = c0 + per < ncol ? c0 + per : ncol;
    double acc = 0.;
    for (int64_t c = c0; c < c1; ++c) acc += rates[c * n + i];
    part[int64_t(blockIdx.y) * n + i] = acc;
}
__global__ void k_reduce_rates_final(int n, const double *__restrict__ part, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc = 0.;
    for (int c = 0; c < RED_CHUNKS; ++c) acc += part[int64_t(c) * n + i];
    out[i] = acc;
}

// The sanity scan the scheme's own 3-D driver runs after every column (M:1025-1094): running maxima of
// qc, qr, nr, qs, qi, qg, ni and a look-out for negative values (there: WARNING strings; here: counts).
// out[0..6] = maxima (>= 0), out[7..14] = number of negative entries of qc,qr,nr,qs,qi,qg,ni,qv.
// Maxima of non-negative doubles order like their bit patterns, so both halves are exact integer atomics.
struct SanityPtrs { const double *v[8]; };
__global__ void k_sanity(int64_t n, SanityPtrs p, unsigned long long *acc)
{
    __shared__ unsigned long long sh[15];
    if (threadIdx.x < 15) sh[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long mx[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned neg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const double x = p.v[a][i];
            if (x < 0.) ++neg[a];
            else if (a < 7 && x > 0.) {                       // +-0 and NaN are no candidates (-0.0 has the largest bit pattern)
                const unsigned long long b = (unsigned long long)__double_as_longlong(x);
                mx[a] = b > mx[a] ? b : mx[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 7; ++a) atomicMax(&sh[a], mx[a]);
#pragma unroll
    for (int a = 0; a < 8; ++a) if (neg[a]) atomicAdd(&sh[7 + a], (unsigned long long)neg[a]);
    __syncthreads();
    if (threadIdx.x < 7) atomicMax(&acc[threadIdx.x], sh[threadIdx.x]);
    else if (threadIdx.x < 15 && sh[threadIdx.x]) atomicAdd(&acc[threadIdx.x], sh[threadIdx.x]);
}
__global__ void k_sanity_final(const unsigned long long *acc, double *out15)
{
    const int i = threadIdx.x;
    if (i < 7) out15[i] = __longlong_as_double((long long)acc[i]);
    else if (i < 15) out15[i] = double(acc[i]);
}

// calc_effectRad, M:4834-4935: effective radii of cloud water, cloud ice and snow for radiation coupling.  Pointwise in
// (column, level); re_* are INOUT (a level without the species keeps the caller's value, M:4873/4888/4897).  The
// reference's column-wide has_qc/has_qi/has_qs flags only skip loops whose bodies test the level again.
struct RadConsts { double Nt_c, cig2, oig1, oams, cse1, sa[10], sb[10]; int aero; };
__global__ void k_effective_radii(int64_t n, RadConsts c, const double *__restrict__ t, const double *__restrict__ p,
                                  const double *__restrict__ qv, const double *__restrict__ qc, const double *__restrict__ nc1,
                                  const double *__restrict__ qi, const double *__restrict__ ni1, const double *__restrict__ qs,
                                  double *__restrict__ re_qc, double *__restrict__ re_qi, double *__restrict__ re_qs)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
#if KFM_TABLES
    fm::tab::load_tables(int(threadIdx.x), int(blockDim.x));           // the snow moment below is a fastmath.h power
    __syncthreads();
#endif
    if (i >= n) return;
    const double am_r_ = PI * rho_w / 6.0, am_i_ = PI * rho_i / 6.0;
    const double rho = 0.622 * p[i] / (Rgas * t[i] * (qv[i] + 0.622));
    const double rc = fmax(R1, qc[i] * rho);
    double nc = fmax(R2, nc1[i] * rho);
    if (!c.aero) nc = c.Nt_c;                                           // .NOT. is_aerosol_aware, M:4863
    const double ri = fmax(R1, qi[i] * rho), ni = fmax(R2, ni1[i] * rho), rs = fmax(R1, qs[i] * rho);
    if (!(rc <= R1 || nc <= R2)) {                                      // M:4873-4884
        int inu_c;
        if (nc < 100.) inu_c = 15;
        else if (nc > 1.E10) inu_c = 2;
        else { inu_c = int(lround(1000.E6 / nc)) + 2; inu_c = inu_c < 15 ? inu_c : 15; }
        const double g_ratio = double((inu_c + 1) * (inu_c + 2) * (inu_c + 3));   // 24, 60, 120 ... 4896 = (n+1)(n+2)(n+3)
        const double lamc = fm::cbrt_pos(nc * am_r_ * g_ratio / rc);
        re_qc[i] = fmax(2.51E-6, fmin(0.5 * double(3. + inu_c) / lamc, 50.E-6));
    }
    if (!(ri <= R1 || ni <= R2)) {                                      // M:4887-4893
        const double lami = fm::cbrt_pos(am_i_ * c.cig2 * c.oig1 * ni / ri);
        re_qi[i] = fmax(5.01E-6, fmin(0.5 * double(3. + mu_i) / lami, 125.E-6));
    }
    if (!(rs <= R1)) {                                                  // M:4896-4930 (bm_s = 2: smo2 = smob)
        const double tc0 = fmin(-0.1, t[i] - 273.15), x = c.cse1;
        const double smob = rs * c.oams;
        const double *a = c.sa, *b = c.sb;
        const double loga_ = a[0] + a[1] * tc0 + a[2] * x + a[3] * tc0 * x + a[4] * tc0 * tc0 + a[5] * x * x
                           + a[6] * tc0 * tc0 * x + a[7] * tc0 * x * x + a[8] * tc0 * tc0 * tc0 + a[9] * x * x * x;
        const double b_ = b[0] + b[1] * tc0 + b[2] * x + b[3] * tc0 * x + b[4] * tc0 * tc0 + b[5] * x * x
                        + b[6] * tc0 * tc0 * x + b[7] * tc0 * x * x + b[8] * tc0 * tc0 * tc0 + b[9] * x * x * x;
        const double smoc = fm::pow10_times_pow(loga_, fm::log2_parts(smob), b_);
        re_qs[i] = fmax(10.E-6, fmin(0.5 * (smoc / smob), 999.E-6));
    }
}

struct Named { const char *name; const double *ptr; int64_t n; };

std::vector<Named> table_dir(const Tables &t)
{
    return {
        {"tcg_racg", t.tcg_racg, N_RACG}, {"tmr_racg", t.tmr_racg, N_RACG}, {"tcr_gacr", t.tcr_gacr, N_RACG},
        {"tmg_gacr", t.tmg_gacr, N_RACG}, {"tnr_racg", t.tnr_racg, N_RACG}, {"tnr_gacr", t.tnr_gacr, N_RACG},
        {"tcs_racs1", t.tcs_racs1, N_RACS}, {"tmr_racs1", t.tmr_racs1, N_RACS}, {"tcs_racs2", t.tcs_racs2, N_RACS},
        {"tmr_racs2", t.tmr_racs2, N_RACS}, {"tcr_sacr1", t.tcr_sacr1, N_RACS}, {"tms_sacr1", t.tms_sacr1, N_RACS},
        {"tcr_sacr2", t.tcr_sacr2, N_RACS}, {"tms_sacr2", t.tms_sacr2, N_RACS}, {"tnr_racs1", t.tnr_racs1, N_RACS},
        {"tnr_racs2", t.tnr_racs2, N_RACS}, {"tnr_sacr1", t.tnr_sacr1, N_RACS}, {"tnr_sacr2", t.tnr_sacr2, N_RACS},
        {"tpi_qcfz", t.tpi_qcfz, N_QCFZ}, {"tni_qcfz", t.tni_qcfz, N_QCFZ},
        {"tpi_qrfz", t.tpi_qrfz, N_QRFZ}, {"tpg_qrfz", t.tpg_qrfz, N_QRFZ}, {"tni_qrfz", t.tni_qrfz, N_QRFZ},
        {"tnr_qrfz", t.tnr_qrfz, N_QRFZ},
        {"tps_iaus", t.tps_iaus, N_IAUS}, {"tni_iaus", t.tni_iaus, N_IAUS}, {"tpi_ide", t.tpi_ide, N_IAUS},
        {"t_Efrw", t.t_Efrw, N_EF}, {"t_Efsw", t.t_Efsw, N_EF}, {"tnc_wev", t.tnc_wev, N_WEV},
        {"racs_rec", t.racs_rec, N_RACS * RACS_REC}, {"racg_rec", t.racg_rec, N_RACG * RACG_REC},
        {"qrfz_rec", t.qrfz_rec, N_QRFZ * QRFZ_REC},
    };
}

std::vector<Named> const_dir(const kidmp_ctx *c)
{
    const Consts &h = c->hc;
    const Bins &b = c->hb;
    return {
        {"Nt_c", &h.Nt_c, 1}, {"Sc3", &h.Sc3, 1}, {"D0i", &h.D0i, 1}, {"xm0s", &h.xm0s, 1}, {"xm0g", &h.xm0g, 1},
        {"cce1", h.cce[0], 15}, {"cce2", h.cce[1], 15}, {"cce3", h.cce[2], 15}, {"cce4", h.cce[3], 15}, {"cce5", h.cce[4], 15},
        {"ccg1", h.ccg[0], 15}, {"ccg2", h.ccg[1], 15}, {"ccg3", h.ccg[2], 15}, {"ccg4", h.ccg[3], 15}, {"ccg5", h.ccg[4], 15},
        {"ocg1", h.ocg1, 15}, {"ocg2", h.ocg2, 15},
        {"cie", h.cie, 7}, {"cig", h.cig, 7}, {"oig1", &h.oig1, 1}, {"oig2", &h.oig2, 1}, {"obmi", &h.obmi, 1},
        {"cre", h.cre, 13}, {"crg", h.crg, 13}, {"ore1", &h.ore1, 1}, {"org1", &h.org1, 1}, {"org2", &h.org2, 1},
        {"org3", &h.org3, 1}, {"obmr", &h.obmr, 1},
        {"cse", h.cse, 18}, {"csg", h.csg, 18}, {"oams", &h.oams, 1}, {"obms", &h.obms, 1}, {"ocms", &h.ocms, 1},
        {"cge", h.cge, 12}, {"cgg", h.cgg, 12}, {"oge1", &h.oge1, 1}, {"ogg1", &h.ogg1, 1}, {"ogg2", &h.ogg2, 1},
        {"ogg3", &h.ogg3, 1}, {"oamg", &h.oamg, 1}, {"obmg", &h.obmg, 1}, {"ocmg", &h.ocmg, 1},
        {"t1_qr_qc", &h.t1_qr_qc, 1}, {"t1_qr_qi", &h.t1_qr_qi, 1}, {"t2_qr_qi", &h.t2_qr_qi, 1},
        {"t1_qg_qc", &h.t1_qg_qc, 1}, {"t1_qs_qc", &h.t1_qs_qc, 1}, {"t1_qs_qi", &h.t1_qs_qi, 1},
        {"t1_qr_ev", &h.t1_qr_ev, 1}, {"t2_qr_ev", &h.t2_qr_ev, 1}, {"t1_qs_sd", &h.t1_qs_sd, 1},
        {"t2_qs_sd", &h.t2_qs_sd, 1}, {"t1_qg_sd", &h.t1_qg_sd, 1}, {"t2_qg_sd", &h.t2_qg_sd, 1},
        {"t1_qs_me", &h.t1_qs_me, 1}, {"t2_qs_me", &h.t2_qs_me, 1}, {"t1_qg_me", &h.t1_qg_me, 1},
        {"t2_qg_me", &h.t2_qg_me, 1},
        {"Dc", b.Dc, nbins}, {"dtc", b.dtc, nbins}, {"Di", b.Di, nbins}, {"dti", b.dti, nbins},
        {"Dr", b.Dr, nbins}, {"dtr", b.dtr, nbins}, {"Ds", b.Ds, nbins}, {"dts", b.dts, nbins},
        {"Dg", b.Dg, nbins}, {"dtg", b.dtg, nbins}, {"t_Nc", b.t_Nc, nbins},
        {"r_c", b.r_c, ntb_c}, {"r_i", b.r_i, ntb_i}, {"r_r", b.r_r, ntb_r}, {"r_g", b.r_g, ntb_g},
        {"r_s", b.r_s, ntb_s}, {"N0r_exp", b.N0r_exp, ntb_r1}, {"N0g_exp", b.N0g_exp, ntb_g1}, {"Nt_i", b.Nt_i, ntb_i1},
    };
}

// device evaluation of the kernel's math helpers (fastmath.h) for the accuracy test
__global__ void k_math_probe(int fn, int64_t n, const double *x, const double *y, double *out)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
#if KFM_TABLES
    fm::tab::load_tables(int(threadIdx.x), int(blockDim.x));
    __syncthreads();
#endif
    if (i >= n) return;
    double r = 0.;
    switch (fn) {
    case KIDMP_MATH_LOG:   r = fm::log(x[i]); break;
    case KIDMP_MATH_LOG10: r = fm::log10(x[i]); break;
    case KIDMP_MATH_EXP:   r = fm::exp(x[i]); break;
    case KIDMP_MATH_EXP10: r = fm::exp10(x[i]); break;
    case KIDMP_MATH_SQRT:  r = fm::sqrt_pos(x[i]); break;
    case KIDMP_MATH_CBRT:  r = fm::cbrt_pos(x[i]); break;
    case KIDMP_MATH_POW:   r = fm::pow(x[i], y[i]); break;
    case 7: r = __builtin_amdgcn_rcp(y[i]); break;                                   // raw v_rcp_f64
    case 8: r = fm::div(x[i], y[i]); break;                                          // the kernel's division
    case 9: r = x[i] / y[i]; break;                                                  // IEEE division (this file is built without -fapprox-func)
    case 10: r = fm::rcp(y[i]); break;                                               // the kernel's reciprocal
    }
    out[i] = r;
}

int check_step_args(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt, const void *const *ptrs, int nptr)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (ncol < 0) return fail(ctx, KIDMP_EINVAL, "kidmp: ncol < 0");
    if (nz < 2 || nz > KIDMP_MAX_NZ) return fail(ctx, KIDMP_EINVAL, "kidmp: nz outside [2, KIDMP_MAX_NZ]");
    if (!(dt > 0.)) return fail(ctx, KIDMP_EINVAL, "kidmp: dt must be > 0");
    for (int i = 0; i < nptr && ncol > 0; ++i)                 // an empty batch has nothing to point at
        if (!ptrs[i]) return fail(ctx, KIDMP_EINVAL, "kidmp: null array argument");
    return KIDMP_OK;
}

// ---- the host-array entries (kidmp_batch_step_host*, kidmp32_batch_step_host): a three-stage pipeline over column chunks ----
// The batch is cut into chunks of CH columns; chunk i is uploaded on the context's H2D stream, stepped on its compute
// stream and downloaded on its D2H stream, through a ring of HOST_NBUF staging sets in HBM, so the two DMA directions
// (PCIe is full duplex) and the kernel work on three different chunks at once.  Host arrays that are page-locked
// (kidmp_host_alloc, or the caller's own hipHostMalloc / hipHostRegister) are moved by the DMA
// engines asynchronously; pageable arrays still work, but the runtime stages them through its own bounce buffer and
// the calling thread waits for each copy.  Per column-step the boundary moves 14 (15 with w) profiles in and 12 out
// (+36 for the rate diagnostics): about 25 KB in binary64.
int64_t pick_host_chunk(const kidmp_ctx *ctx, int64_t ncol)
{
    if (ctx->host_chunk > 0) return ctx->host_chunk < ncol ? ctx->host_chunk : ncol;
    if (ncol <= 2048) return ncol;                            // one chunk: nothing to overlap with
    int64_t ch = (ncol + 3) / 4;                              // at least four chunks ...
    ch = (ch + 255) / 256 * 256;
    return ch > 8192 ? 8192 : ch;                             // ... of at most 8 192 columns (7.9 MB per profile slice; measured optimum)
}

// Leaving host_pipeline with an error must not leave DMA in flight towards the caller's arrays.
struct PipelineDrain {
    kidmp_ctx *c;
    bool armed = true;
    ~PipelineDrain()
    {
        if (!armed) return;
        (void)hipStreamSynchronize(c->s_h2d);
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->s_d2h);
    }
};

template <class T, class Launch>
int host_pipeline(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt, T *const *io, const T *const *in, T *ppt,
                  double *rates, int32_t *nstep, Launch launch, bool exact_sums = false, bool scan_sanity = false)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    // Arrays the caller may leave out (NULL), as KiD itself does (W:36 passes nc1d, nwfa1d, nifa1d unset; a warm run
    // never touches the frozen species, W:46-52): they then neither cross PCIe nor come back.
    //   nc, nwfa, nifa (all three)   non-aerosol contexts: the defaults of M:958-964, formed on the device
    //   qi, qs, qg, ni (all four)    iiwarm contexts: exactly zero (and they stay zero)
    const bool has_w = ctx->cfg.is_aerosol_aware != 0;
    const int n_aer = (io[8] != nullptr) + (io[9] != nullptr) + (io[10] != nullptr);
    const int n_frz = (io[2] != nullptr) + (io[4] != nullptr) + (io[5] != nullptr) + (io[6] != nullptr);
    const bool skip_aer = n_aer == 0 && ncol > 0, skip_frz = n_frz == 0 && ncol > 0;
    if (ncol > 0 && n_aer != 0 && n_aer != 3) return fail(ctx, KIDMP_EINVAL, "kidmp: nc, nwfa, nifa must be given or left out together");
    if (ncol > 0 && n_frz != 0 && n_frz != 4) return fail(ctx, KIDMP_EINVAL, "kidmp: qi, qs, qg, ni must be given or left out together");
    if (skip_aer && has_w) return fail(ctx, KIDMP_EINVAL, "kidmp: an aerosol-aware context needs nc, nwfa and nifa");
    if (skip_frz && !ctx->cfg.iiwarm) return fail(ctx, KIDMP_EINVAL, "kidmp: a mixed-phase context needs qi, qs, qg and ni");
    const void *ptrs[15];
    int np = 0;
    for (int v = 0; v < 12; ++v) {
        const bool optional_out = (skip_aer && v >= 8 && v <= 10) || (skip_frz && (v == 2 || v == 4 || v == 5 || v == 6));
        if (!optional_out) ptrs[np++] = io[v];
    }
    ptrs[np++] = in[0]; ptrs[np++] = in[1]; ptrs[np++] = ppt;
    if (int rc = check_step_args(ctx, ncol, nz, dt, ptrs, np)) return rc;
    if (has_w && !in[2] && ncol > 0) return fail(ctx, KIDMP_EINVAL, "kidmp: an aerosol-aware context needs the updraft profile w");
    if (ncol == 0) {
        if (exact_sums || scan_sanity) {
            GUARD(ctx);
            if (exact_sums) HIPTRY(ctx, hipMemset(ctx->d_acc, 0, ACC_N * sizeof(unsigned long long)));
            if (scan_sanity) HIPTRY(ctx, hipMemset(ctx->d_sanity, 0, 15 * sizeof(unsigned long long)));
        }
        return KIDMP_OK;
    }
    GUARD(ctx);
    const int64_t CH = pick_host_chunk(ctx, ncol);
    const int64_t nchunk = (ncol + CH - 1) / CH;
    const int nbuf = nchunk < HOST_NBUF ? int(nchunk) : HOST_NBUF;
    const size_t prof = size_t(CH) * size_t(nz);
    // one staging set: [rates (double)] [15 profiles + ppt (T)] [nstep (int32)], each part 256-byte aligned
    auto up256 = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t b_rates = rates ? up256(size_t(KIDMP_NRATES) * prof * sizeof(double)) : 0;
    const size_t b_prof = up256(prof * sizeof(T));
    const size_t b_ppt = up256(4 * size_t(CH) * sizeof(T));
    const size_t b_nstep = nstep ? up256(4 * size_t(CH) * sizeof(int32_t)) : 0;
    const size_t b_set = b_rates + 15 * b_prof + b_ppt + b_nstep;
    const size_t need = b_set * size_t(nbuf);
    if (need > ctx->stage_bytes) {
        if (ctx->d_stage) (void)hipFree(ctx->d_stage);
        ctx->d_stage = nullptr;
        ctx->stage_bytes = 0;
        HIPTRY(ctx, hipMalloc((void **)&ctx->d_stage, need));
        ctx->stage_bytes = need;
    }
    char *const base = reinterpret_cast<char *>(ctx->d_stage);
    PipelineDrain drain{ctx};
    if (exact_sums) HIPTRY(ctx, hipMemsetAsync(ctx->d_acc, 0, ACC_N * sizeof(unsigned long long), ctx->stream));
    if (scan_sanity) HIPTRY(ctx, hipMemsetAsync(ctx->d_sanity, 0, 15 * sizeof(unsigned long long), ctx->stream));
    for (int64_t i = 0; i < nchunk; ++i) {
        const int b = int(i % nbuf);
        const int64_t c0 = i * CH, n = (c0 + CH <= ncol ? CH : ncol - c0);
        const size_t off = size_t(c0) * size_t(nz), cnt = size_t(n) * size_t(nz);
        char *set = base + size_t(b) * b_set;
        double *drates = rates ? reinterpret_cast<double *>(set) : nullptr;
        T *dio[12]; const T *din[3];
        char *q = set + b_rates;
        for (int v = 0; v < 12; ++v) { dio[v] = reinterpret_cast<T *>(q); q += b_prof; }
        T *dinw[3];
        for (int v = 0; v < 3; ++v) { dinw[v] = reinterpret_cast<T *>(q); din[v] = dinw[v]; q += b_prof; }
        T *dppt = reinterpret_cast<T *>(q); q += b_ppt;
        int32_t *dnstep = nstep ? reinterpret_cast<int32_t *>(q) : nullptr;
        if (!has_w || !in[2]) din[2] = nullptr;
        // upload (the set is free once the download of the chunk that used it last has finished)
        if (i >= nbuf) HIPTRY(ctx, hipStreamWaitEvent(ctx->s_h2d, ctx->ev_down[b], 0));
        for (int v = 0; v < 12; ++v)
            if (io[v]) HIPTRY(ctx, hipMemcpyAsync(dio[v], io[v] + off, cnt * sizeof(T), hipMemcpyHostToDevice, ctx->s_h2d));
        for (int v = 0; v < (din[2] ? 3 : 2); ++v) HIPTRY(ctx, hipMemcpyAsync(dinw[v], in[v] + off, cnt * sizeof(T), hipMemcpyHostToDevice, ctx->s_h2d));
        HIPTRY(ctx, hipMemcpyAsync(dppt, ppt + 4 * c0, 4 * size_t(n) * sizeof(T), hipMemcpyHostToDevice, ctx->s_h2d));
        HIPTRY(ctx, hipEventRecord(ctx->ev_up[b], ctx->s_h2d));
        // step
        HIPTRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_up[b], 0));
        if (skip_frz)
            for (int v : {2, 4, 5, 6}) HIPTRY(ctx, hipMemsetAsync(dio[v], 0, cnt * sizeof(T), ctx->stream));
        if (skip_aer) {
            hipLaunchKernelGGL(k_default_aerosols<T>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, int64_t(cnt),
                               T(ctx->hc.Nt_c), dio[0], dio[11], din[0], dio[8], dio[9], dio[10]);
            HIPTRY(ctx, hipGetLastError());
        }
        if (int rc = launch(n, dio, din, dppt, drates, dnstep)) return rc;
        if (exact_sums) HIPTRY(ctx, launch_ppt_exact<T>(n, dppt, ctx->d_acc, ctx->stream));   // the chunk's share of the domain sums
        if constexpr (std::is_same<T, double>::value)
            if (scan_sanity) {                                // the scan of M:1025-1094 over the chunk's end state (exact integer atomics)
                SanityPtrs sp{{dio[1], dio[3], dio[7], dio[4], dio[2], dio[5], dio[6], dio[0]}};
                int64_t g = (int64_t(cnt) + 255) / 256;
                if (g > 2048) g = 2048;
                hipLaunchKernelGGL(k_sanity, dim3((unsigned)g), dim3(256), 0, ctx->stream, int64_t(cnt), sp, ctx->d_sanity);
                HIPTRY(ctx, hipGetLastError());
            }
        HIPTRY(ctx, hipEventRecord(ctx->ev_step[b], ctx->stream));
        // download
        HIPTRY(ctx, hipStreamWaitEvent(ctx->s_d2h, ctx->ev_step[b], 0));
        for (int v = 0; v < 12; ++v)
            if (io[v]) HIPTRY(ctx, hipMemcpyAsync(io[v] + off, dio[v], cnt * sizeof(T), hipMemcpyDeviceToHost, ctx->s_d2h));
        HIPTRY(ctx, hipMemcpyAsync(ppt + 4 * c0, dppt, 4 * size_t(n) * sizeof(T), hipMemcpyDeviceToHost, ctx->s_d2h));
        if (rates) HIPTRY(ctx, hipMemcpyAsync(rates + size_t(KIDMP_NRATES) * off, drates, size_t(KIDMP_NRATES) * cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->s_d2h));
        if (nstep) HIPTRY(ctx, hipMemcpyAsync(nstep + 4 * c0, dnstep, 4 * size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->s_d2h));
        HIPTRY(ctx, hipEventRecord(ctx->ev_down[b], ctx->s_d2h));
    }
    HIPTRY(ctx, hipStreamSynchronize(ctx->s_d2h));           // everything else precedes it through the events
    drain.armed = false;
    return KIDMP_OK;
}

struct CacheFamily { const char *file; std::vector<double *> dev; int64_t n; };
std::vector<CacheFamily> cache_families(Tables &t)
{
    return {
        {"racg_thompson09.data", {t.tcg_racg, t.tmr_racg, t.tcr_gacr, t.tmg_gacr, t.tnr_racg, t.tnr_gacr}, N_RACG},   // M:3823-3828
        {"racs_thompson09.data", {t.tcs_racs1, t.tmr_racs1, t.tcs_racs2, t.tmr_racs2, t.tcr_sacr1, t.tms_sacr1,
                                  t.tcr_sacr2, t.tms_sacr2, t.tnr_racs1, t.tnr_racs2, t.tnr_sacr1, t.tnr_sacr2}, N_RACS},   // M:4066-4077
    };
}

}  // namespace

extern "C" {

int kidmp_init(const kidmp_cfg *cfg, kidmp_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, KIDMP_EINVAL, "kidmp_init: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, KIDMP_ENODEV, "kidmp_init: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, KIDMP_ENODEV, "kidmp_init: bad device ordinal");
    if (!(cfg->set_Nc > 0.)) return fail(nullptr, KIDMP_EINVAL, "kidmp_init: set_Nc must be > 0");
    kidmp_ctx *c = new (std::nothrow) kidmp_ctx;
    if (!c) return fail(nullptr, KIDMP_ENOMEM, "kidmp_init: out of host memory");
    c->cfg = *cfg;
#ifdef KIDMP_PROFILING
    if (const char *e = getenv("KIDMP_DEBUG_STOP")) c->debug_stop = atoi(e);   // libkidmp_prof.so only: truncates the step
#endif
    const auto t0 = std::chrono::steady_clock::now();
    auto bail = [&](int code) { kidmp_finalize(c); return code; };
    DeviceGuard guard_(cfg->device);                 // the caller's current device is restored on every exit path
    {
        if (guard_.err != hipSuccess) { g_err = std::string("hipSetDevice: ") + hipGetErrorString(guard_.err); return bail(KIDMP_EHIP); }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && !strstr(prop.gcnArchName, "gfx950")) {
            g_err = std::string("kidmp_init: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only";
            return bail(KIDMP_ENODEV);
        }
    }
    host_init(cfg->iiwarm ? 1 : 0, cfg->l_sediment ? 1 : 0, cfg->set_Nc, c->hc, c->hb);
    if (!generated_consts_match(c->hc)) {
        g_err = "kidmp_init: thompson_consts_gen.h is stale (rebuild: make -C kid_amd/csrc clean all)";
        return bail(KIDMP_ESTATE);
    }
    hipError_t e;
#define INITTRY(x) do { e = (x); if (e != hipSuccess) { g_err = std::string(#x ": ") + hipGetErrorString(e); return bail(KIDMP_EHIP); } } while (0)
    INITTRY(hipStreamCreate(&c->stream));
    INITTRY(hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
    INITTRY(hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking));
    for (int b = 0; b < HOST_NBUF; ++b) {
        INITTRY(hipEventCreateWithFlags(&c->ev_up[b], hipEventDisableTiming));
        INITTRY(hipEventCreateWithFlags(&c->ev_step[b], hipEventDisableTiming));
        INITTRY(hipEventCreateWithFlags(&c->ev_down[b], hipEventDisableTiming));
    }
    // scratch of the diagnostics entries, sized for KIDMP_MAX_NZ once: no entry allocates after kidmp_init
    c->red_elems = size_t(RED_CHUNKS) * size_t(KIDMP_NRATES) * size_t(KIDMP_MAX_NZ);
    INITTRY(hipMalloc((void **)&c->d_red, c->red_elems * sizeof(double)));
    INITTRY(hipMalloc((void **)&c->d_sanity, 15 * sizeof(unsigned long long)));
    INITTRY(hipMalloc((void **)&c->d_acc, ACC_N * sizeof(unsigned long long)));
    INITTRY(hipMalloc((void **)&c->d_consts, sizeof(Consts)));
    INITTRY(hipMalloc((void **)&c->d_bins, sizeof(Bins)));
    INITTRY(hipMemcpy(c->d_consts, &c->hc, sizeof(Consts), hipMemcpyHostToDevice));
    INITTRY(hipMemcpy(c->d_bins, &c->hb, sizeof(Bins), hipMemcpyHostToDevice));
    c->cslot = take_slot();
    if (c->cslot < 0) { g_err = "kidmp_init: more than 8 live contexts in this process"; return bail(KIDMP_ESTATE); }
    INITTRY(p64::upload_consts(c->cslot, c->hc));       // one constant-memory image per arithmetic variant
    INITTRY(p32n::upload_consts(c->cslot, c->hc));
    INITTRY(f32::upload_consts(c->cslot, c->hc));
    INITTRY(alloc_tables(c->tables));
    INITTRY(build_tables(c->d_consts, c->d_bins, c->hc.iiwarm, c->tables, c->stream));
#undef INITTRY
    c->init_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->ready = true;
    *out = c;
    return KIDMP_OK;
}

void kidmp_finalize(kidmp_ctx *c)
{
    if (!c) return;
    DeviceGuard guard_(c->cfg.device);
    free_tables(c->tables);
    if (c->d_consts) (void)hipFree(c->d_consts);
    if (c->d_bins) (void)hipFree(c->d_bins);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->d_sanity) (void)hipFree(c->d_sanity);
    if (c->d_acc) (void)hipFree(c->d_acc);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d);
    if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h);
    for (int b = 0; b < HOST_NBUF; ++b) {
        if (c->ev_up[b]) (void)hipEventDestroy(c->ev_up[b]);
        if (c->ev_step[b]) (void)hipEventDestroy(c->ev_step[b]);
        if (c->ev_down[b]) (void)hipEventDestroy(c->ev_down[b]);
    }
    give_slot(c->cslot);
    delete c;
}

const char *kidmp_last_error(const kidmp_ctx *c) { return c && !c->err.empty() ? c->err.c_str() : g_err.c_str(); }
double kidmp_init_seconds(const kidmp_ctx *c) { return c ? c->init_s : 0.; }
const char *kidmp_kernel_name(void) { return column_kernel_name(); }

int kidmp_batch_step_device(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt,
                            double *qv, double *qc, double *qi, double *qr, double *qs, double *qg,
                            double *ni, double *nr, double *nc, double *nwfa, double *nifa, double *t,
                            const double *p, const double *w, const double *dz,
                            double *ppt, double *rates, int32_t *nstep, void *stream)
{
    // w1d only feeds activ_ncloud (is_aerosol_aware, M:2797): optional otherwise
    const void *ptrs[] = {qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, dz, ppt};
    if (int rc = check_step_args(ctx, ncol, nz, dt, ptrs, 15)) return rc;
    if (ctx->cfg.is_aerosol_aware && !w) return fail(ctx, KIDMP_EINVAL, "kidmp: an aerosol-aware context needs the updraft profile w");
    GUARD(ctx);
    if (int rc = check_on_device(ctx, qv, "qv")) return rc;
    if (int rc = check_on_device(ctx, ppt, "ppt")) return rc;
    StepArgs a{};
    a.qv = qv; a.qc = qc; a.qi = qi; a.qr = qr; a.qs = qs; a.qg = qg; a.ni = ni; a.nr = nr;
    a.nc = nc; a.nwfa = nwfa; a.nifa = nifa; a.t = t; a.p = p; a.dz = dz;
    a.ppt = ppt; a.rates = rates; a.nstep = nstep;
    a.cslot = ctx->cslot; a.tables = ctx->tables; a.iiwarm = ctx->cfg.iiwarm != 0;
    a.aero = ctx->cfg.is_aerosol_aware != 0; a.w = w;
    a.ncol = ncol; a.nz = nz; a.dt = dt;
    a.debug_stop = ctx->debug_stop;
    if (ncol == 0) return KIDMP_OK;
    HIPTRY(ctx, p64::launch_column_step(a, (hipStream_t)stream));
    return KIDMP_OK;
}

void *kidmp_host_alloc(size_t bytes)
{
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) { g_err = std::string("kidmp_host_alloc: ") + hipGetErrorString(e); return nullptr; }
    return p;
}
void kidmp_host_free(void *p) { if (p) (void)hipHostFree(p); }
int kidmp_set_host_chunk(kidmp_ctx *ctx, int64_t ncol_per_chunk)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (ncol_per_chunk < 0) return fail(ctx, KIDMP_EINVAL, "kidmp_set_host_chunk: negative chunk size");
    ctx->host_chunk = ncol_per_chunk;
    return KIDMP_OK;
}

int kidmp_batch_step_host(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt,
                          double *qv, double *qc, double *qi, double *qr, double *qs, double *qg,
                          double *ni, double *nr, double *nc, double *nwfa, double *nifa, double *t,
                          const double *p, const double *w, const double *dz, double *ppt, double *rates)
{
    return kidmp_batch_step_host_diag(ctx, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz,
                                      ppt, rates, nullptr);
}

int kidmp_batch_step_host_diag(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt,
                               double *qv, double *qc, double *qi, double *qr, double *qs, double *qg,
                               double *ni, double *nr, double *nc, double *nwfa, double *nifa, double *t,
                               const double *p, const double *w, const double *dz, double *ppt, double *rates,
                               int32_t *nstep)
{
    double *io[12] = {qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t};
    const double *in[3] = {p, dz, w};
    return host_pipeline<double>(ctx, ncol, nz, dt, io, in, ppt, rates, nstep,
        [&](int64_t n, double *const *d, const double *const *f, double *dppt, double *drates, int32_t *dnstep) {
            return kidmp_batch_step_device(ctx, n, nz, dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11],
                                           f[0], f[2], f[1], dppt, drates, dnstep, ctx->stream);
        });
}

int kidmp_column_step(kidmp_ctx *ctx, int32_t nz, double dt,
                      double *qv1d, double *qc1d, double *qi1d, double *qr1d, double *qs1d, double *qg1d,
                      double *ni1d, double *nr1d, double *nc1d, double *nwfa1d, double *nifa1d, double *t1d,
                      const double *p1d, const double *w1d, const double *dzq, double *ppt)
{
    return kidmp_batch_step_host(ctx, 1, nz, dt, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d,
                                 nifa1d, t1d, p1d, w1d, dzq, ppt, nullptr);
}

// ---- binary32 state: the reference as shipped (P32n) and the all-binary32 build ----
int kidmp32_batch_step_device(kidmp_ctx *ctx, int64_t ncol, int32_t nz, float dt,
                              float *qv, float *qc, float *qi, float *qr, float *qs, float *qg,
                              float *ni, float *nr, float *nc, float *nwfa, float *nifa, float *t,
                              const float *p, const float *w, const float *dz,
                              float *ppt, double *rates, int32_t *nstep, int32_t arith, void *stream)
{
    const void *ptrs[] = {qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, dz, ppt};
    if (int rc = check_step_args(ctx, ncol, nz, double(dt), ptrs, 15)) return rc;
    if (ctx->cfg.is_aerosol_aware && !w) return fail(ctx, KIDMP_EINVAL, "kidmp: an aerosol-aware context needs the updraft profile w");
    if (arith != KIDMP_ARITH_P32N && arith != KIDMP_ARITH_F32) return fail(ctx, KIDMP_EINVAL, "kidmp32: arith must be KIDMP_ARITH_P32N or KIDMP_ARITH_F32");
    GUARD(ctx);
    if (int rc = check_on_device(ctx, qv, "qv")) return rc;
    if (int rc = check_on_device(ctx, ppt, "ppt")) return rc;
    StepArgsT<float> a{};
    a.qv = qv; a.qc = qc; a.qi = qi; a.qr = qr; a.qs = qs; a.qg = qg; a.ni = ni; a.nr = nr;
    a.nc = nc; a.nwfa = nwfa; a.nifa = nifa; a.t = t; a.p = p; a.dz = dz;
    a.ppt = ppt; a.rates = rates; a.nstep = nstep;
    a.cslot = ctx->cslot; a.tables = ctx->tables; a.iiwarm = ctx->cfg.iiwarm != 0;
    a.aero = ctx->cfg.is_aerosol_aware != 0; a.w = w;
    a.ncol = ncol; a.nz = nz; a.dt = dt;
    a.debug_stop = ctx->debug_stop;
    if (ncol == 0) return KIDMP_OK;
    if (arith == KIDMP_ARITH_P32N) HIPTRY(ctx, p32n::launch_column_step(a, (hipStream_t)stream));
    else                           HIPTRY(ctx, f32::launch_column_step(a, (hipStream_t)stream));
    return KIDMP_OK;
}

int kidmp32_batch_step_host(kidmp_ctx *ctx, int64_t ncol, int32_t nz, float dt,
                            float *qv, float *qc, float *qi, float *qr, float *qs, float *qg,
                            float *ni, float *nr, float *nc, float *nwfa, float *nifa, float *t,
                            const float *p, const float *w, const float *dz, float *ppt, double *rates,
                            int32_t *nstep, int32_t arith)
{
    float *io[12] = {qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t};
    const float *in[3] = {p, dz, w};
    if (arith != KIDMP_ARITH_P32N && arith != KIDMP_ARITH_F32) return fail(ctx, KIDMP_EINVAL, "kidmp32: arith must be KIDMP_ARITH_P32N or KIDMP_ARITH_F32");
    return host_pipeline<float>(ctx, ncol, nz, double(dt), io, in, ppt, rates, nstep,
        [&](int64_t n, float *const *d, const float *const *f, float *dppt, double *drates, int32_t *dnstep) {
            return kidmp32_batch_step_device(ctx, n, nz, dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11],
                                             f[0], f[2], f[1], dppt, drates, dnstep, arith, ctx->stream);
        });
}

int kidmp32_column_step(kidmp_ctx *ctx, int32_t nz, float dt,
                        float *qv1d, float *qc1d, float *qi1d, float *qr1d, float *qs1d, float *qg1d,
                        float *ni1d, float *nr1d, float *nc1d, float *nwfa1d, float *nifa1d, float *t1d,
                        const float *p1d, const float *w1d, const float *dzq, float *ppt, int32_t arith)
{
    return kidmp32_batch_step_host(ctx, 1, nz, dt, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d,
                                   nifa1d, t1d, p1d, w1d, dzq, ppt, nullptr, nullptr, arith);
}

int kidmp_default_aerosols_device(kidmp_ctx *ctx, int64_t n, const double *qv, const double *t, const double *p,
                                  double *nc, double *nwfa, double *nifa, void *stream)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (n < 0 || !qv || !t || !p || !nc || !nwfa || !nifa) return fail(ctx, KIDMP_EINVAL, "kidmp_default_aerosols_device: bad argument");
    if (n == 0) return KIDMP_OK;
    GUARD(ctx);
    if (int rc = check_on_device(ctx, qv, "qv")) return rc;
    if (int rc = check_on_device(ctx, nc, "nc")) return rc;
    const int T = 256;
    hipLaunchKernelGGL(k_default_aerosols<double>, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, (hipStream_t)stream, n,
                       ctx->hc.Nt_c, qv, t, p, nc, nwfa, nifa);
    HIPTRY(ctx, hipGetLastError());
    return KIDMP_OK;
}

int kidmp_math_probe(kidmp_ctx *ctx, int32_t fn, int64_t n, const double *x, const double *y, double *out)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (n < 0 || !x || !y || !out || fn < 0 || fn > 10) return fail(ctx, KIDMP_EINVAL, "kidmp_math_probe: bad argument");
    if (n == 0) return KIDMP_OK;
    GUARD(ctx);
    double *d = nullptr;
    HIPTRY(ctx, hipMalloc(&d, size_t(n) * 3 * sizeof(double)));
    hipError_t e = hipMemcpy(d, x, size_t(n) * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, y, size_t(n) * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, fn, n, d, d + n, d + 2 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, size_t(n) * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIPTRY(ctx, e);
    return KIDMP_OK;
}

int kidmp_reduce_ppt_device(kidmp_ctx *ctx, int64_t ncol, const double *ppt, double *out4, void *stream)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (ncol < 0 || !ppt || !out4) return fail(ctx, KIDMP_EINVAL, "kidmp_reduce_ppt_device: bad argument");
    GUARD(ctx);
    if (int rc = check_on_device(ctx, ppt, "ppt")) return rc;
    if (int rc = check_on_device(ctx, out4, "out4")) return rc;
    hipLaunchKernelGGL(k_reduce_ppt, dim3(1), dim3(256), 0, (hipStream_t)stream, ncol, ppt, out4);
    HIPTRY(ctx, hipGetLastError());
    return KIDMP_OK;
}

int kidmp_reduce_rates_device(kidmp_ctx *ctx, int64_t ncol, int32_t nz, const double *rates, double *out, void *stream)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (ncol < 0 || nz < 2 || nz > KIDMP_MAX_NZ || !rates || !out) return fail(ctx, KIDMP_EINVAL, "kidmp_reduce_rates_device: bad argument");
    GUARD(ctx);
    if (int rc = check_on_device(ctx, rates, "rates")) return rc;
    if (int rc = check_on_device(ctx, out, "out")) return rc;
    const int n = KIDMP_NRATES * nz;
    if (size_t(RED_CHUNKS) * size_t(n) > ctx->red_elems) return fail(ctx, KIDMP_EINVAL, "kidmp_reduce_rates_device: nz beyond KIDMP_MAX_NZ");
    const int T = 128;
    hipLaunchKernelGGL(k_reduce_rates_part, dim3((n + T - 1) / T, RED_CHUNKS), dim3(T), 0, (hipStream_t)stream, ncol, n, rates, ctx->d_red);
    hipLaunchKernelGGL(k_reduce_rates_final, dim3((n + T - 1) / T), dim3(T), 0, (hipStream_t)stream, n, ctx->d_red, out);
    HIPTRY(ctx, hipGetLastError());
    return KIDMP_OK;
}

int kidmp_sanity_device(kidmp_ctx *ctx, int64_t n, const double *qc, const double *qr, const double *nr, const double *qs,
                        const double *qi, const double *qg, const double *ni, const double *qv, double *out15, void *stream)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (n < 0 || !qc || !qr || !nr || !qs || !qi || !qg || !ni || !qv || !out15) return fail(ctx, KIDMP_EINVAL, "kidmp_sanity_device: bad argument");
    GUARD(ctx);
    if (int rc = check_on_device(ctx, qc, "qc")) return rc;
    if (int rc = check_on_device(ctx, out15, "out15")) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIPTRY(ctx, hipMemsetAsync(ctx->d_sanity, 0, 15 * sizeof(unsigned long long), s));
    if (n > 0) {
        SanityPtrs p{{qc, qr, nr, qs, qi, qg, ni, qv}};
        const int T = 256;
        int64_t g = (n + T - 1) / T;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(k_sanity, dim3((unsigned)g), dim3(T), 0, s, n, p, ctx->d_sanity);
    }
    hipLaunchKernelGGL(k_sanity_final, dim3(1), dim3(64), 0, s, ctx->d_sanity, out15);
    HIPTRY(ctx, hipGetLastError());
    return KIDMP_OK;
}

int kidmp_effective_radii_device(kidmp_ctx *ctx, int64_t n, const double *t, const double *p, const double *qv,
                                 const double *qc, const double *nc, const double *qi, const double *ni, const double *qs,
                                 double *re_qc, double *re_qi, double *re_qs, void *stream)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (n < 0 || !t || !p || !qv || !qc || !nc || !qi || !ni || !qs || !re_qc || !re_qi || !re_qs)
        return fail(ctx, KIDMP_EINVAL, "kidmp_effective_radii_device: bad argument");
    if (n == 0) return KIDMP_OK;
    GUARD(ctx);
    if (int rc = check_on_device(ctx, t, "t")) return rc;
    if (int rc = check_on_device(ctx, re_qc, "re_qc")) return rc;
    RadConsts c{};
    c.aero = ctx->cfg.is_aerosol_aware != 0;
    c.Nt_c = ctx->hc.Nt_c; c.cig2 = ctx->hc.cig[1]; c.oig1 = ctx->hc.oig1; c.oams = ctx->hc.oams; c.cse1 = ctx->hc.cse[0];
    for (int i = 0; i < 10; ++i) { c.sa[i] = ctx->hc.sa[i]; c.sb[i] = ctx->hc.sb[i]; }
    const int T = 256;
    hipLaunchKernelGGL(k_effective_radii, dim3((unsigned)((n + T - 1) / T)), dim3(T), 0, (hipStream_t)stream, n, c,
                       t, p, qv, qc, nc, qi, ni, qs, re_qc, re_qi, re_qs);
    HIPTRY(ctx, hipGetLastError());
    return KIDMP_OK;
}

const char *kidmp_kernel_fingerprint(kidmp_ctx *ctx)
{
    if (!ctx || !ctx->ready) return "";
    DeviceGuard guard_(ctx->cfg.device);
    ctx->fingerprint = p64::column_kernel_fingerprint(ctx->cfg.iiwarm != 0);
    return ctx->fingerprint.c_str();
}

const char *kidmp32_kernel_fingerprint(kidmp_ctx *ctx, int32_t arith)
{
    if (!ctx || !ctx->ready || (arith != KIDMP_ARITH_P32N && arith != KIDMP_ARITH_F32)) return "";
    DeviceGuard guard_(ctx->cfg.device);
    ctx->fingerprint = arith == KIDMP_ARITH_P32N ? p32n::column_kernel_fingerprint(ctx->cfg.iiwarm != 0)
                                                 : f32::column_kernel_fingerprint(ctx->cfg.iiwarm != 0);
    return ctx->fingerprint.c_str();
}

int64_t kidmp_get_table(kidmp_ctx *ctx, const char *name, double *out, int64_t cap)
{
    if (!ctx || !ctx->ready || !name) return fail(ctx, KIDMP_ESTATE, "kidmp_get_table: bad context");
    GUARD(ctx);
    for (const Named &e : table_dir(ctx->tables))
        if (!strcmp(e.name, name)) {
            if (!out) return e.n;
            if (cap < e.n) return fail(ctx, KIDMP_EINVAL, "kidmp_get_table: buffer too small");
            HIPTRY(ctx, hipMemcpy(out, e.ptr, size_t(e.n) * sizeof(double), hipMemcpyDeviceToHost));
            return e.n;
        }
    return fail(ctx, KIDMP_EINVAL, std::string("kidmp_get_table: unknown table ") + name);
}

int64_t kidmp_get_const(kidmp_ctx *ctx, const char *name, double *out, int64_t cap)
{
    if (!ctx || !ctx->ready || !name) return fail(ctx, KIDMP_ESTATE, "kidmp_get_const: bad context");
    for (const Named &e : const_dir(ctx))
        if (!strcmp(e.name, name)) {
            if (!out) return e.n;
            if (cap < e.n) return fail(ctx, KIDMP_EINVAL, "kidmp_get_const: buffer too small");
            memcpy(out, e.ptr, size_t(e.n) * sizeof(double));
            return e.n;
        }
    return fail(ctx, KIDMP_EINVAL, std::string("kidmp_get_const: unknown constant ") + name);
}

int kidmp_cache_write_file(const char *path, int32_t ntab, const double *const *tabs, int64_t n_each)
{
    if (!path || !tabs || ntab <= 0 || n_each <= 0) return fail(nullptr, KIDMP_EINVAL, "kidmp_cache_write_file: bad argument");
    return cache_write(path, ntab, tabs, n_each) == 0 ? KIDMP_OK : fail(nullptr, KIDMP_EIO, std::string("cannot write ") + path);
}

int kidmp_cache_read_file(const char *path, int32_t ntab, double *const *tabs, int64_t n_each)
{
    if (!path || !tabs || ntab <= 0 || n_each <= 0) return fail(nullptr, KIDMP_EINVAL, "kidmp_cache_read_file: bad argument");
    const int rc = cache_read(path, ntab, tabs, n_each);
    if (rc == -1) return fail(nullptr, KIDMP_EIO, std::string("cannot open ") + path);
    if (rc != 0) return fail(nullptr, KIDMP_EIO, std::string("malformed or short table cache ") + path);
    return KIDMP_OK;
}

int kidmp_save_table_cache(kidmp_ctx *ctx, const char *dir)
{
    if (!ctx || !ctx->ready || !dir) return fail(ctx, KIDMP_ESTATE, "kidmp_save_table_cache: bad context");
    if (ctx->hc.iiwarm) return fail(ctx, KIDMP_ESTATE, "kidmp_save_table_cache: iiwarm context has no mixed-phase tables");
    GUARD(ctx);
    for (CacheFamily &fam : cache_families(ctx->tables)) {
        std::vector<std::vector<double>> host(fam.dev.size(), std::vector<double>(size_t(fam.n)));
        std::vector<const double *> ptr;
        for (size_t i = 0; i < fam.dev.size(); ++i) {
            HIPTRY(ctx, hipMemcpy(host[i].data(), fam.dev[i], size_t(fam.n) * sizeof(double), hipMemcpyDeviceToHost));
            ptr.push_back(host[i].data());
        }
        const std::string path = std::string(dir) + "/" + fam.file;
        if (cache_write(path.c_str(), int(ptr.size()), ptr.data(), fam.n) != 0) return fail(ctx, KIDMP_EIO, "cannot write " + path);
    }
    return KIDMP_OK;
}

// thompson_init's use of the cache files, per file as in the reference: qr_acr_qg (M:3717-3729, M:3822-3829) and
// qr_acr_qs (M:3864-3895, M:4065-4078) each do
//     inquire(file=..., exist=fexist);  fexist = fexist .and. l_reuse_thompson_lookup
//     if (fexist) then  read the 6 (12) tables  else  compute them and write(12,*) / write(13,*) them
int kidmp_table_cache_reuse(kidmp_ctx *ctx, const char *dir, int32_t l_reuse, int32_t write_if_built, int32_t *status)
{
    if (status) *status = 0;
    if (!ctx || !ctx->ready || !dir) return fail(ctx, KIDMP_ESTATE, "kidmp_table_cache_reuse: bad context");
    if (ctx->hc.iiwarm) return KIDMP_OK;                     // thompson_init builds these tables only if .not. iiwarm (M:773)
    GUARD(ctx);
    struct stat sb;
    const bool have_dir = stat(dir, &sb) == 0 && S_ISDIR(sb.st_mode);
    bool loaded = false;
    int fam_no = 0;
    for (CacheFamily &fam : cache_families(ctx->tables)) {
        const std::string path = std::string(dir) + "/" + fam.file;
        bool fexist = false;
        if (FILE *f = std::fopen(path.c_str(), "r")) { fexist = true; std::fclose(f); }
        std::vector<std::vector<double>> host(fam.dev.size(), std::vector<double>(size_t(fam.n)));
        if (fexist && l_reuse) {
            std::vector<double *> ptr;
            for (auto &h : host) ptr.push_back(h.data());
            const int rc = cache_read(path.c_str(), int(ptr.size()), ptr.data(), fam.n);
            if (rc != 0) return fail(ctx, KIDMP_EIO, (rc == -1 ? "cannot open " : "malformed or short table cache ") + path);
            for (size_t i = 0; i < fam.dev.size(); ++i)
                HIPTRY(ctx, hipMemcpy(fam.dev[i], host[i].data(), size_t(fam.n) * sizeof(double), hipMemcpyHostToDevice));
            loaded = true;
            if (status) *status |= 1 << fam_no;
        } else if (write_if_built) {
            // the reference opens the file unconditionally and aborts without the directory (M:3718); here a missing
            // directory just means nothing is written (reported through *status)
            if (have_dir) {
                std::vector<const double *> ptr;
                for (size_t i = 0; i < fam.dev.size(); ++i) {
                    HIPTRY(ctx, hipMemcpy(host[i].data(), fam.dev[i], size_t(fam.n) * sizeof(double), hipMemcpyDeviceToHost));
                    ptr.push_back(host[i].data());
                }
                if (cache_write(path.c_str(), int(ptr.size()), ptr.data(), fam.n) != 0) return fail(ctx, KIDMP_EIO, "cannot write " + path);
                if (status) *status |= 4 << fam_no;
            }
        }
        ++fam_no;
    }
    if (loaded) HIPTRY(ctx, repack_records(ctx->tables, ctx->stream));   // the solver reads the interleaved records
    return KIDMP_OK;
}

int kidmp_load_table_cache(kidmp_ctx *ctx, const char *dir)
{
    if (!ctx || !ctx->ready || !dir) return fail(ctx, KIDMP_ESTATE, "kidmp_load_table_cache: bad context");
    if (ctx->hc.iiwarm) return fail(ctx, KIDMP_ESTATE, "kidmp_load_table_cache: iiwarm context has no mixed-phase tables");
    GUARD(ctx);
    for (CacheFamily &fam : cache_families(ctx->tables)) {
        std::vector<std::vector<double>> host(fam.dev.size(), std::vector<double>(size_t(fam.n)));
        std::vector<double *> ptr;
        for (auto &h : host) ptr.push_back(h.data());
        const std::string path = std::string(dir) + "/" + fam.file;
        const int rc = cache_read(path.c_str(), int(ptr.size()), ptr.data(), fam.n);
        if (rc != 0) return fail(ctx, KIDMP_EIO, (rc == -1 ? "cannot open " : "malformed or short table cache ") + path);
        for (size_t i = 0; i < fam.dev.size(); ++i)
            HIPTRY(ctx, hipMemcpy(fam.dev[i], host[i].data(), size_t(fam.n) * sizeof(double), hipMemcpyHostToDevice));
    }
    HIPTRY(ctx, repack_records(ctx->tables, ctx->stream));       // the solver reads the interleaved records
    return KIDMP_OK;
}

}  // extern "C"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Several GPUs behind one call: what a Fortran / C host (KiD's `do i=1,nx`, W:54-246, with nx in the millions) reaches
// without MPI.  Columns are independent and the tables read-only, so the batch is cut into contiguous ranges, one per
// context (= per device), each range goes through that context's own upload / step / download pipeline on its own host
// thread, and the ONE exchange of the path -- the domain sums of the surface precipitation, the nx-means of W:248-303
// -- is an RCCL all-reduce over the devices of the exact integer accumulators (k_ppt_exact): 24 int64, SUM.
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;

const char *load_rccl()       // nullptr on success, else what failed
{
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_rccl.lib) return nullptr;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return "librccl.so not found (dlopen)";
    RcclApi a;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString)
        return "librccl.so lacks an expected symbol";
    a.lib = h;
    g_rccl = a;
    return nullptr;
}

}  // namespace

struct kidmp_multi {
    std::vector<kidmp_ctx *> ctx;            // one per entry of the device list, in list order
    std::vector<int> leader;                 // contexts that lead a distinct device (entries may repeat a device)
    std::vector<int> leader_of;              // ctx index -> index into `leader`
    std::vector<ncclComm_t> comm;            // one RCCL communicator per distinct device
    std::string err;
};

namespace {

int mfail(kidmp_multi *m, int code, const std::string &msg)
{
    if (m) m->err = msg;
    g_err = msg;
    return code;
}

// the 24 limbs -> four doubles: carries propagated in 128-bit integers, then the digits summed from the top in long
// double (64-bit significand): a pure function of the limbs, so equal limbs give equal sums
void limbs_to_sums(const int64_t *limbs, double *out4)
{
    for (int sp = 0; sp < 4; ++sp) {
        __int128 carry = 0;
        long double v = 0.0L;
        long double digit[ACC_LIMBS + 1];
        for (int j = 0; j < ACC_LIMBS; ++j) {
            const __int128 t = (__int128)limbs[sp * ACC_LIMBS + j] + carry;
            const __int128 lowbits = t & (__int128)0xffffffffLL;           // 0 .. 2**32-1
            carry = (t - lowbits) >> 32;                                   // exact: t - lowbits is a multiple of 2**32
            digit[j] = (long double)(int64_t)lowbits;
        }
        digit[ACC_LIMBS] = (long double)(int64_t)carry;                    // signed top
        for (int j = ACC_LIMBS; j >= 0; --j) v += __builtin_ldexpl(digit[j], 32 * j - 128);
        out4[sp] = (double)v;
    }
}

}  // namespace

extern "C" {

int kidmp_shard_bounds(int64_t ncol, int32_t nshard, int32_t shard, int64_t *lo, int64_t *hi)
{
    if (ncol < 0 || nshard < 1 || shard < 0 || shard >= nshard || !lo || !hi) return fail(nullptr, KIDMP_EINVAL, "kidmp_shard_bounds: bad argument");
    const int64_t base = ncol / nshard, rem = ncol % nshard;               // contiguous ranges, sizes differ by at most one
    *lo = shard * base + (shard < rem ? shard : rem);
    *hi = *lo + base + (shard < rem ? 1 : 0);
    return KIDMP_OK;
}

int kidmp_ppt_limbs_to_sums(const int64_t *limbs, double *out4)
{
    if (!limbs || !out4) return fail(nullptr, KIDMP_EINVAL, "kidmp_ppt_limbs_to_sums: null argument");
    limbs_to_sums(limbs, out4);
    return KIDMP_OK;
}

int kidmp_reduce_ppt_exact_device(kidmp_ctx *ctx, int64_t ncol, const double *ppt, int64_t *limbs, void *stream)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (ncol < 0 || !limbs || (ncol > 0 && !ppt)) return fail(ctx, KIDMP_EINVAL, "kidmp_reduce_ppt_exact_device: bad argument");
    GUARD(ctx);
    if (int rc = check_on_device(ctx, ppt, "ppt")) return rc;
    if (int rc = check_on_device(ctx, limbs, "limbs")) return rc;
    HIPTRY(ctx, hipMemsetAsync(limbs, 0, ACC_N * sizeof(int64_t), (hipStream_t)stream));
    HIPTRY(ctx, launch_ppt_exact<double>(ncol, ppt, reinterpret_cast<unsigned long long *>(limbs), (hipStream_t)stream));
    return KIDMP_OK;
}

void kidmp_finalize_multi(kidmp_multi *m)
{
    if (!m) return;
    for (ncclComm_t c : m->comm)
        if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    for (kidmp_ctx *c : m->ctx) kidmp_finalize(c);
    delete m;
}

int kidmp_init_multi(const kidmp_cfg *cfg, int32_t ndev, const int32_t *devices, kidmp_multi **out)
{
    if (!cfg || !out || !devices || ndev < 1 || ndev > KIDMP_MAX_DEVICE_LIST) return fail(nullptr, KIDMP_EINVAL, "kidmp_init_multi: bad argument (1..16 devices)");
    *out = nullptr;
    kidmp_multi *m = new (std::nothrow) kidmp_multi;
    if (!m) return fail(nullptr, KIDMP_ENOMEM, "kidmp_init_multi: out of host memory");
    for (int i = 0; i < ndev; ++i) {
        kidmp_cfg c = *cfg;
        c.device = devices[i];
        kidmp_ctx *x = nullptr;
        const int rc = kidmp_init(&c, &x);
        if (rc != KIDMP_OK) { kidmp_finalize_multi(m); return rc; }          // message already in g_err
        m->ctx.push_back(x);
        int l = -1;
        for (size_t q = 0; q < m->leader.size(); ++q)
            if (m->ctx[m->leader[q]]->cfg.device == devices[i]) l = int(q);
        if (l < 0) { m->leader.push_back(i); l = int(m->leader.size()) - 1; }
        m->leader_of.push_back(l);
    }
    // RCCL: one communicator per DISTINCT device (a list may name a device twice -- two contexts sharing a card, which
    // is how a one-GPU box exercises this path; their accumulators are added before the collective)
    if (const char *why = load_rccl()) { kidmp_finalize_multi(m); return fail(nullptr, KIDMP_ENODEV, std::string("kidmp_init_multi: ") + why); }
    std::vector<int> devs;
    for (int l : m->leader) devs.push_back(m->ctx[l]->cfg.device);
    m->comm.assign(devs.size(), nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), int(devs.size()), devs.data());
    if (r != ncclSuccess) {
        const std::string msg = std::string("kidmp_init_multi: ncclCommInitAll: ") + g_rccl.GetErrorString(r);
        for (auto &c : m->comm) c = nullptr;
        kidmp_finalize_multi(m);
        return fail(nullptr, KIDMP_EHIP, msg);
    }
    *out = m;
    return KIDMP_OK;
}

int32_t kidmp_multi_size(const kidmp_multi *m) { return m ? int32_t(m->ctx.size()) : 0; }
kidmp_ctx *kidmp_multi_context(kidmp_multi *m, int32_t i) { return m && i >= 0 && size_t(i) < m->ctx.size() ? m->ctx[size_t(i)] : nullptr; }
const char *kidmp_multi_last_error(const kidmp_multi *m) { return m && !m->err.empty() ? m->err.c_str() : g_err.c_str(); }

int kidmp_batch_step_host_multi_diag(kidmp_multi *m, int64_t ncol, int32_t nz, double dt,
                                     double *qv, double *qc, double *qi, double *qr, double *qs, double *qg,
                                     double *ni, double *nr, double *nc, double *nwfa, double *nifa, double *t,
                                     const double *p, const double *w, const double *dz, double *ppt, double *rates,
                                     int32_t *nstep, double *precip_sums, double *sanity15)
{
    if (!m || m->ctx.empty()) return mfail(m, KIDMP_ESTATE, "kidmp_batch_step_host_multi: not initialised");
    if (ncol < 0 || nz < 2 || nz > KIDMP_MAX_NZ) return mfail(m, KIDMP_EINVAL, "kidmp_batch_step_host_multi: bad ncol / nz");
    const int nctx = int(m->ctx.size());
    // Nothing below may throw through the C boundary: allocation failures (std::bad_alloc from the vectors, std::system_error
    // from std::thread) are mapped to a status code, and threads that did start are joined before the function returns.
    std::vector<int> rc;
    std::vector<std::string> msg;
    std::vector<int64_t> limbs, lead;
    std::vector<unsigned long long> san, san_lead;
    std::vector<std::thread> th;
    try {
        rc.assign(size_t(nctx), KIDMP_OK);
        msg.resize(size_t(nctx));
        limbs.resize(size_t(nctx) * ACC_N);
        lead.assign(m->leader.size() * ACC_N, 0);
        san.resize(size_t(nctx) * 15);
        san_lead.assign(m->leader.size() * 15, 0ull);
        th.reserve(size_t(nctx));
    } catch (const std::exception &) {
        return mfail(m, KIDMP_ENOMEM, "kidmp_batch_step_host_multi: out of host memory");
    }
    auto work = [&](int i) noexcept {
        try {
            int64_t lo = 0, hi = 0;
            kidmp_shard_bounds(ncol, nctx, i, &lo, &hi);
            const size_t o = size_t(lo) * size_t(nz);
            auto at = [o](double *a) { return a ? a + o : nullptr; };
            auto atc = [o](const double *a) { return a ? a + o : nullptr; };
            kidmp_ctx *c = m->ctx[size_t(i)];
            double *io[12] = {at(qv), at(qc), at(qi), at(qr), at(qs), at(qg), at(ni), at(nr), at(nc), at(nwfa), at(nifa), at(t)};
            const double *in[3] = {atc(p), atc(dz), atc(w)};
            rc[size_t(i)] = host_pipeline<double>(c, hi - lo, nz, dt, io, in, ppt ? ppt + 4 * lo : nullptr,
                rates ? rates + size_t(KIDMP_NRATES) * o : nullptr, nstep ? nstep + 4 * lo : nullptr,
                [&](int64_t n, double *const *d, const double *const *f, double *dppt, double *drates, int32_t *dnstep) {
                    return kidmp_batch_step_device(c, n, nz, dt, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11],
                                                   f[0], f[2], f[1], dppt, drates, dnstep, c->stream);
                }, true, sanity15 != nullptr);
            if (rc[size_t(i)] != KIDMP_OK) msg[size_t(i)] = kidmp_last_error(c);
        } catch (const std::bad_alloc &) {
            rc[size_t(i)] = KIDMP_ENOMEM;
        } catch (...) {
            rc[size_t(i)] = KIDMP_EHIP;
        }
    };
    // one host thread per context: HIP's current device and the pipeline's blocking waits are per thread
    int started = 0;
    bool thread_failure = false;
    for (int i = 1; i < nctx; ++i) {
        try {
            th.emplace_back(work, i);
            ++started;
        } catch (const std::exception &) {                     // std::system_error: no more threads
            thread_failure = true;
            break;
        }
    }
    if (!thread_failure) work(0);
    for (auto &x : th) x.join();
    if (thread_failure)
        return mfail(m, KIDMP_ENOMEM, "kidmp_batch_step_host_multi: could not start a host thread per context (" +
                                      std::to_string(started) + " of " + std::to_string(nctx - 1) + " started, joined; nothing was stepped on the others)");
    for (int i = 0; i < nctx; ++i)
        if (rc[size_t(i)] != KIDMP_OK)
            return mfail(m, rc[size_t(i)], "device " + std::to_string(m->ctx[size_t(i)]->cfg.device) + ": " +
                                           (msg[size_t(i)].empty() ? std::string("host-side failure in the context's worker thread") : msg[size_t(i)]));
    if (!precip_sums && !sanity15) return KIDMP_OK;
    // ---- the domain diagnostics: contexts that share a device combine their accumulators on the host, then the devices
    //      exchange them: all-reduce(int64, SUM) of the 24 precipitation limbs and -- on request, the analogue of the scan
    //      of M:1025-1094 -- all-reduce(uint64, MAX) of the 7 maxima (bit patterns of non-negative doubles order like the
    //      values) and all-reduce(uint64, SUM) of the 8 negative-entry counts, in ONE RCCL group ----
    // A leader's stream must be idle before this function returns on ANY path (queued collectives / copies).
    struct DrainLeaders {
        kidmp_multi *m;
        ~DrainLeaders()
        {
            for (int l : m->leader) {
                kidmp_ctx *c = m->ctx[size_t(l)];
                DeviceGuard g(c->cfg.device);
                (void)hipStreamSynchronize(c->stream);
            }
        }
    } drain_leaders{m};
    for (int i = 0; i < nctx; ++i) {
        kidmp_ctx *c = m->ctx[size_t(i)];
        DeviceGuard g(c->cfg.device);
        hipError_t e = hipMemcpy(&limbs[size_t(i) * ACC_N], c->d_acc, ACC_N * sizeof(int64_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess && sanity15) e = hipMemcpy(&san[size_t(i) * 15], c->d_sanity, 15 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return mfail(m, KIDMP_EHIP, std::string("hipMemcpy(accumulators): ") + hipGetErrorString(e));
        const size_t L = size_t(m->leader_of[size_t(i)]);
        for (int q = 0; q < ACC_N; ++q)                                    // wrap-around addition == two's complement sum
            lead[L * ACC_N + q] = int64_t(uint64_t(lead[L * ACC_N + q]) + uint64_t(limbs[size_t(i) * ACC_N + q]));
        if (sanity15)
            for (int q = 0; q < 15; ++q) {
                const unsigned long long v = san[size_t(i) * 15 + q];
                san_lead[L * 15 + q] = q < 7 ? (v > san_lead[L * 15 + q] ? v : san_lead[L * 15 + q]) : san_lead[L * 15 + q] + v;
            }
    }
    for (size_t l = 0; l < m->leader.size(); ++l) {                        // 192 + 120 bytes per device: synchronous copies,
        kidmp_ctx *c = m->ctx[size_t(m->leader[l])];                       // so that no DMA ever reads a host buffer after this scope
        DeviceGuard g(c->cfg.device);
        hipError_t e = hipMemcpy(c->d_acc, &lead[l * ACC_N], ACC_N * sizeof(int64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess && sanity15) e = hipMemcpy(c->d_sanity, &san_lead[l * 15], 15 * sizeof(unsigned long long), hipMemcpyHostToDevice);
        if (e != hipSuccess) return mfail(m, KIDMP_EHIP, std::string("hipMemcpy(accumulators, to device): ") + hipGetErrorString(e));
    }
    ncclResult_t r = g_rccl.GroupStart();
    for (size_t l = 0; l < m->leader.size() && r == ncclSuccess; ++l) {
        kidmp_ctx *c = m->ctx[size_t(m->leader[l])];
        DeviceGuard g(c->cfg.device);
        r = g_rccl.AllReduce(c->d_acc, c->d_acc, ACC_N, ncclInt64, ncclSum, m->comm[l], c->stream);
        if (r == ncclSuccess && sanity15) r = g_rccl.AllReduce(c->d_sanity, c->d_sanity, 7, ncclUint64, ncclMax, m->comm[l], c->stream);
        if (r == ncclSuccess && sanity15) r = g_rccl.AllReduce(c->d_sanity + 7, c->d_sanity + 7, 8, ncclUint64, ncclSum, m->comm[l], c->stream);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return mfail(m, KIDMP_EHIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    int64_t total[ACC_N];
    unsigned long long stot[15];
    for (size_t l = 0; l < m->leader.size(); ++l) {                        // every device holds the same results; all are drained
        kidmp_ctx *c = m->ctx[size_t(m->leader[l])];
        DeviceGuard g(c->cfg.device);
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess && l == 0) e = hipMemcpy(total, c->d_acc, sizeof(total), hipMemcpyDeviceToHost);
        if (e == hipSuccess && l == 0 && sanity15) e = hipMemcpy(stot, c->d_sanity, sizeof(stot), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return mfail(m, KIDMP_EHIP, std::string("all-reduce of the domain diagnostics: ") + hipGetErrorString(e));
    }
    if (precip_sums) limbs_to_sums(total, precip_sums);
    if (sanity15)
        for (int q = 0; q < 15; ++q) {
            double v;
            if (q < 7) memcpy(&v, &stot[q], sizeof(v)); else v = double(stot[q]);
            sanity15[q] = v;
        }
    return KIDMP_OK;
}

int kidmp_batch_step_host_multi(kidmp_multi *m, int64_t ncol, int32_t nz, double dt,
                                double *qv, double *qc, double *qi, double *qr, double *qs, double *qg,
                                double *ni, double *nr, double *nc, double *nwfa, double *nifa, double *t,
                                const double *p, const double *w, const double *dz, double *ppt, double *rates,
                                int32_t *nstep, double *precip_sums)
{
    return kidmp_batch_step_host_multi_diag(m, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt, rates,
                                            nstep, precip_sums, nullptr);
}

/* deprecated: device memory is no longer reserved per batch (launches own no per-batch memory since round 2); kept so that
 * hosts linked against earlier builds keep linking.  Checks its arguments and does nothing. */
int kidmp_reserve(kidmp_ctx *ctx, int64_t ncol, int32_t nz)
{
    if (!ctx || !ctx->ready) return fail(ctx, KIDMP_ESTATE, "kidmp: context not initialised");
    if (ncol < 0 || nz < 2 || nz > KIDMP_MAX_NZ) return fail(ctx, KIDMP_EINVAL, "kidmp_reserve: bad ncol / nz");
    return KIDMP_OK;
}

}  // extern "C"
