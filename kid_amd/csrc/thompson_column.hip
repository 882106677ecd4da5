// thompson_column.hip -- the Thompson-09n column step for gfx950 (MI355X).
//
// Replaces subroutine mp_thompson (M:1156-3688 of the reference's
// module_mp_thompson09n.f90) for batches of independent columns.
//
// Mapping: ONE WAVEFRONT PER COLUMN, lanes over the vertical index.
//   * KiD stores a profile as nz contiguous reals (theta(k,i), W:60), so a
//     wave reads/writes a profile as one contiguous burst; consecutive
//     columns are consecutive in memory, so the grid's loads are contiguous
//     over the column index as well.
//   * lane l owns levels k = l + 64*j, j < NJ (NJ = 2 for nz = 120).
//   * the column's per-level state that must survive a phase boundary is
//     staged in LDS ([slot][k], conflict-free ds_read/write_b64); the ~100
//     values that are live inside a phase stay in VGPRs.
//   * the few vertical couplings of the scheme are wave-level primitives:
//       - no_micro (M:1396..1521)            ballot
//       - k_0 = top level with T>=270.65     max-reduce      (M:1634-1637)
//       - graupel N0 running minimum         suffix-min scan (M:1638-1654)
//       - fall speed "carry down"            suffix propagate scan (M:3235...)
//       - nstep, ksed1                       max-reduce      (M:3239-3246)
//       - upwind flux sed(k+1)               lane shift      (M:3381...)
//   * lookup tables stay in HBM/Infinity Cache; the rain-snow / rain-graupel /
//     rain-freezing families are read as interleaved per-cell records.
//   * a workgroup is four waves = four consecutive columns.  The pointwise
//     rate sweep (blocks D-N, 60 % of the instructions) is shared out by
//     ALTITUDE BAND instead: one wave pass = 16 levels of all four columns,
//     so the lanes of a pass sit in one microphysical regime (see the kernel).
//   * two instantiations: mixed phase (256 VGPRs, 21 LDS slots, 2 waves per
//     SIMD) and warm rain (iiwarm contexts: frozen-species blocks compiled
//     out, 168 VGPRs, 13 slots, 3 waves per SIMD).
// No MFMA: pointwise transcendental rates plus a vertical sweep.
//
// Arithmetic: fp64 throughout (the reference's P64 build), compiled with
// -ffp-contract=off so products and sums round as in the Fortran.  libm calls
// are fastmath.h (range-specific, <= ~2 ulp); division is rcp + Newton (<= 1 ulp).
//
// Reference UB given defined semantics (same decisions as the oracle):
//   U1 cloud water does not sediment (vtck/vtnck never assigned, M:3414-3425).
//   U4 the t_Efrw/t_Efsw droplet-size index is clamped to the table.
#include <hip/hip_runtime.h>

#include "thompson_column.h"
#include "fastmath.h"
#include "thompson_consts_gen.h"   // kc::*, generated at build time by gen_consts.cpp

#include <cstdlib>

namespace kidmp {

namespace {

constexpr int WAVE = 64;

// ---------------- wave primitives ----------------
// Cross-lane work stays on the VALU: DPP row shifts inside the four 16-lane rows, v_readlane to
// stitch rows (and the NJ level groups) together.  No LDS round trips (ds_bpermute) on these paths.
__device__ inline int lane_id() { return threadIdx.x & (WAVE - 1); }

constexpr int DPP_ROW_SHL = 0x100;      // row_shl:n  lane i <- lane i+n of its 16-lane row
constexpr int DPP_WAVE_SHL1 = 0x130;    // wave_shl:1 lane i <- lane i+1 across the wave (gfx9 family)

template <int CTRL>
__device__ inline int dpp_i(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);   // lanes without a source keep `old`
}
template <int CTRL>
__device__ inline double dpp_d(double old, double v)
{
    const long long ov = __double_as_longlong(old), vv = __double_as_longlong(v);
    const unsigned lo = unsigned(dpp_i<CTRL>(int(unsigned(ov)), int(unsigned(vv))));
    const unsigned hi = unsigned(dpp_i<CTRL>(int(unsigned(ov >> 32)), int(unsigned(vv >> 32))));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ inline double readlane_d(double v, int l)
{
    const long long vv = __double_as_longlong(v);
    const unsigned lo = unsigned(__builtin_amdgcn_readlane(int(unsigned(vv)), l));
    const unsigned hi = unsigned(__builtin_amdgcn_readlane(int(unsigned(vv >> 32)), l));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// max over the wave of a non-negative int (result uniform)
__device__ inline int wave_max_i(int v)
{
    v = max(v, dpp_i<DPP_ROW_SHL + 1>(0, v));
    v = max(v, dpp_i<DPP_ROW_SHL + 2>(0, v));
    v = max(v, dpp_i<DPP_ROW_SHL + 4>(0, v));
    v = max(v, dpp_i<DPP_ROW_SHL + 8>(0, v));                            // lanes 0,16,32,48 hold their row's max
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16),
              r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

// highest level lane + 64 j whose predicate holds (0 if none); uniform, from the ballots
template <int NJ>
__device__ inline int top_level(const bool (&p)[NJ])
{
    int ks = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const unsigned long long m = __ballot(p[j]);
        if (m) ks = 63 - __clzll((long long)m) + WAVE * j;
    }
    return ks;
}

// x[j] <- min over all levels at or above (lane + 64 j)   (suffix scan from the column top)
template <int NJ>
__device__ inline void suffix_min(double (&x)[NJ])
{
    const int row = lane_id() >> 4;
    const double inf = __builtin_inf();
    double carry = inf;                                                  // min over the level groups above
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
        double v = x[j];
        v = fmin(v, dpp_d<DPP_ROW_SHL + 1>(inf, v));
        v = fmin(v, dpp_d<DPP_ROW_SHL + 2>(inf, v));
        v = fmin(v, dpp_d<DPP_ROW_SHL + 4>(inf, v));
        v = fmin(v, dpp_d<DPP_ROW_SHL + 8>(inf, v));                     // suffix min inside each row
        const double t3 = fmin(readlane_d(v, 48), carry);                // rows 3.., incl. carry
        const double t2 = fmin(readlane_d(v, 32), t3);
        const double t1 = fmin(readlane_d(v, 16), t2);
        const double above = row == 3 ? carry : (row == 2 ? t3 : (row == 1 ? t2 : t1));
        v = fmin(v, above);
        x[j] = v;
        carry = fmin(readlane_d(v, 0), carry);
    }
}

// a[j] (and b[j]) <- value at the nearest level at or above that has ok, else 0
// (the "vtXk(k) = vtXk(k+1)" carry of M:3235, 3267, 3307, 3333).
// The nearest valid level is found arithmetically from the ballot of `ok` (first set bit at or above the own
// lane) and fetched with one bpermute per 32-bit half; groups are chained from the top through lane 0.
__device__ inline double bperm_d(double v, int src_lane)
{
    const long long vv = __double_as_longlong(v);
    const unsigned lo = unsigned(__builtin_amdgcn_ds_bpermute(src_lane << 2, int(unsigned(vv))));
    const unsigned hi = unsigned(__builtin_amdgcn_ds_bpermute(src_lane << 2, int(unsigned(vv >> 32))));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int NJ, bool TWO>
__device__ inline void carry_down(double (&a)[NJ], double (&b)[NJ], const int (&okin)[NJ])
{
    const int lane = lane_id();
    double ca = 0., cb = 0.;                             // value carried in from the groups above (0: none)
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
        const unsigned long long m = __ballot(okin[j] != 0) >> lane;     // bit i: lane+i is valid
        const bool found = m != 0ull;
        const int src = found ? lane + __ffsll((long long)m) - 1 : lane;
        const double fa = bperm_d(a[j], src);
        a[j] = found ? fa : ca;
        if (TWO) {
            const double fb = bperm_d(b[j], src);
            b[j] = found ? fb : cb;
        }
        ca = readlane_d(a[j], 0);
        if (TWO) cb = readlane_d(b[j], 0);
    }
}

// sed(k+1) for every owned level (0 above the top slot)
template <int NJ>
__device__ inline void shift_from_above(const double (&s)[NJ], double (&up)[NJ])
{
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const double nxt = j + 1 < NJ ? readlane_d(s[j + 1 < NJ ? j + 1 : j], 0) : 0.;   // lane 63 <- next group
        up[j] = dpp_d<DPP_WAVE_SHL1>(nxt, s[j]);
    }
}


// Per-group values (arr[j], j < NJ) that are touched inside a non-unrolled loop over j: a run-time subscript
// would force the array onto the scratch stack, so the element is chosen by compare/select instead.
template <class T, int NJ>
__device__ inline T pick(const T (&arr)[NJ], int j)
{
    T v = arr[0];
#pragma unroll
    for (int i = 1; i < NJ; ++i) v = (j == i) ? arr[i] : v;
    return v;
}
template <class T, int NJ>
__device__ inline void put(T (&arr)[NJ], int j, T v)
{
#pragma unroll
    for (int i = 0; i < NJ; ++i) arr[i] = (j == i) ? v : arr[i];
}

// ---- powers with exponents fixed by the scheme's PARAMETERs ----
// The reference writes x**cre(n) etc. with run-time exponents, but every one of
// them is a compile-time constant of the scheme (M:452-553 from bm_*, bv_*, mu_*).
// Where that constant is an integer, a half-integer, 1/3, 1/4 or 1/6 the power is
// taken with multiplies, sqrt and cbrt (each correctly rounded or <= 1 ulp), which
// agrees with libm pow to 1-3 ulp at a fraction of its ~300 fp64 instructions.
static_assert(bm_r == 3.0 && bm_i == 3.0 && bm_g == 3.0 && bm_s == 2.0, "mass exponents");
static_assert(mu_r == 0.0 && mu_g == 0.0 && mu_i == 0.0 && bv_r == 1.0 && bv_i == 1.0, "PSD/fallspeed exponents");
__device__ inline double cube(double x) { return x * x * x; }                      // **bm_r, **bm_i, **bm_g
__device__ inline double pw4(double x) { const double s = x * x; return s * s; }   // **cre(1), cre(3), cre(9), cge(1)
__device__ inline double pw5(double x) { const double s = x * x; return s * s * x; }           // **cre(6)
__device__ inline double pw7(double x) { const double s = x * x; return s * s * s * x; }       // **cre(8)
__device__ inline double pw2h(double x) { return x * x * fm::sqrt_pos(x); }                // **cre(12) = 2.5
__device__ inline double pw3h(double x) { return x * x * x * fm::sqrt_pos(x); }            // **cre(7)  = 3.5
__device__ inline double root3(double x) { return fm::cbrt_pos(x); }                       // **obmr, **obmi, **obmg
__device__ inline double root4(double x) { return fm::sqrt_pos(fm::sqrt_pos(x)); }                 // **oge1 = 1/(bm_g+1)
__device__ inline double root6(double x) { return fm::sqrt_pos(fm::cbrt_pos(x)); }                 // **(1./6.), M:1701

// ---- log / exp / general powers: fastmath.h (argument-range-specific, ~2 ulp, about half the instructions
//      of the device math library) ----
using fm::Log2Parts;
using fm::log2_parts;
using fm::pow10_times_pow;
__device__ inline double fpow(double x, double y) { return fm::pow(x, y); }

// ---------------- scalar helpers ----------------
// Decade index of M:1763-1771 and its seven siblings:
//     nic = NINT(ALOG10(x));  n = first of {nic-1, nic, nic+1} with x/10.**n in [1,10);
//     idx = INT(x/10.**n) + 10*(n-n0) - (n-n0), clamped to 1..ntb
// where 10.**n is real**integer (compiler-rt __powidf2: the exact power 10^|n| by squaring, then a
// reciprocal for n < 0).  Exactly one n has a mantissa in [1,10), so it is found here without the
// log10: estimate n from the binary exponent, build 10^|n| exactly from its bits (every partial
// product is a power of ten <= 1e22, hence exact, hence equal to __powidf2's), form the same two
// divisions as the reference and step n if the mantissa falls outside [1,10).
__device__ inline double pow10_abs(int m)                // 10^m for 0 <= m <= 31 (exact up to 22)
{
    double t = (m & 1) ? 10. : 1.;
    t *= (m & 2) ? 100. : 1.;
    t *= (m & 4) ? 1.e4 : 1.;
    t *= (m & 8) ? 1.e8 : 1.;
    t *= (m & 16) ? 1.e16 : 1.;
    return t;
}
__device__ inline int decade_idx(double x, int n0, int ntb)
{
    const int e2 = int((__double_as_longlong(x) >> 52) & 0x7ff) - 1023;
    int n = (e2 * 1233) >> 12;                           // ~ floor(e2*log10(2)), off by at most one
    double q = 1.;
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        const double T = pow10_abs(n < 0 ? -n : n);
        const double P = n < 0 ? 1. / T : T;             // 10.**n
        q = x / P;
        if (q >= 10.0) ++n;
        else if (q < 1.0) --n;
        else break;
    }
    int idx = int(q) + 10 * (n - n0) - (n - n0);
    idx = idx < ntb ? idx : ntb;
    return idx > 1 ? idx : 1;
}

// Flatau et al. saturation mixing ratios, M:4656-4717
__device__ inline double rslf(double P, double T)
{
    const double X = fmax(-80., T - 273.16);
    double e = .611583699E03 + X * (.444606896E02 + X * (.143177157E01 + X * (.264224321E-1 + X * (.299291081E-3
             + X * (.203154182E-5 + X * (.702620698E-8 + X * (.379534310E-11 + X * -.321582393E-13)))))));
    e = fmin(e, P * 0.15);
    return .622 * e / (P - e);
}
__device__ inline double rsif(double P, double T)
{
    const double X = fmax(-80., T - 273.16);
    double e = .609868993E03 + X * (.499320233E02 + X * (.184672631E01 + X * (.402737184E-1 + X * (.565392987E-3
             + X * (.521693933E-5 + X * (.307839583E-7 + X * (.105785160E-9 + X * .161444444E-12)))))));
    e = fmin(e, P * 0.15);
    return .622 * e / (P - e);
}

// diffu = 2.11E-5*(T/273.15)**1.94*(101325./p), M:1522.  T/273.15 lies in [0.6,1.3], so
// |1.94 ln x| < 1 and exp(1.94*log(x)) carries the rounding of log/exp straight through
// (<= 3 ulp from libm pow) at a third of pow's cost.
__device__ inline double diffusivity(double temp, double pres)
{
    return 2.11E-5 * fm::exp(1.94 * fm::log(temp / 273.15)) * (101325. / pres);
}

__device__ inline double visc_air(double tempc)          // M:1524-1528
{
    return tempc >= 0.0 ? (1.718 + 0.0049 * tempc) * 1.0E-5
                        : (1.718 + 0.0049 * tempc - 1.2E-5 * tempc * tempc) * 1.0E-5;
}

// Field et al. (2005) fits: sum in the source's term order (M:1590-1599)
__device__ inline double fit(const double *s, double tc, double x)
{
    return s[0] + s[1] * tc + s[2] * x + s[3] * tc * x + s[4] * tc * tc + s[5] * x * x
         + s[6] * tc * tc * x + s[7] * tc * x * x + s[8] * tc * tc * tc + s[9] * x * x * x;
}
__device__ inline double snow_moment(const Log2Parts &lsmo2, double tc0, double order)
{
    // a_ = 10.**loga_ ; moment = a_ * smo2**b_   (M:1595-1600)
    return pow10_times_pow(fit(kc::sa, tc0, order), lsmo2, fit(kc::sb, tc0, order));
}

// graupel intercept before the running minimum, M:1639-1647
__device__ inline double graupel_N0(bool use_rain, double mvd_r, double rg)
{
    const double xslw1 = use_rain ? 4.01 + fm::log10(mvd_r) : 0.01;
    const double ygra1 = 4.31 + fm::log10(fmax(5.E-5, rg));
    const double zans1 = 3.1 + (100. / (300. * xslw1 * ygra1 / (10. / xslw1 + 1. + 0.25 * ygra1) + 30. + 10. * ygra1));
    const double N0 = fm::exp10(zans1);
    return fmax(gonv_min, fmin(N0, gonv_max));
}

// rain number from a prescribed median volume diameter (M:1453-1454 and siblings)
__device__ inline double nr_from_mvd(const Consts &c, double rr, double mvd)
{
    const double lamr = (3.0 + mu_r + 0.672) / mvd;
    return kc::crg[1] * kc::org3 * rr * cube(lamr) / am_r;
}

// LDS slots ([slot][level]).  S0 = after block C, S1 = after block J, S2 = after block N.
// A level's slots are only ever touched by the lane that owns the level, and each
// phase writes its results after it has read its inputs, so later phases reuse slots.
enum Slot {
    // S0 (written in pass 0, read in pass 1)
    V_TEMP = 0, V_QV, V_RHO, V_RC, V_RI, V_RR, V_RS, V_RG, V_NI, V_NR, V_QVSI, V_SSATW, V_SSATI, V_DIFFU, V_N0X,
    //    raw inputs that pass 1 needs as well: fetched once, with the rest of the column, in pass 0
    V_PRES, V_NWFA, V_NIFA, V_NCRAW, V_NIRAW, V_NRRAW,
    // S1 (written at the end of pass 1)
    V_TTEN = 0, V_QVTEN, V_QCTEN, V_NCTEN, V_QITEN, V_NITEN, V_QRTEN, V_NRTEN, V_QSTEN, V_QGTEN,
    V_PRRGML, V_BOOST,
    // S2 (written at the end of pass 2): keeps the tendencies, drops qvten / prr_gml
    V_OCP = V_QVTEN, V_LVAP = V_PRRGML,
    V_TEMP2 = 12, V_RHO2, V_RI2, V_NI2, V_RR2, V_NR2, V_RS2, V_RG2, V_VTS0,
    NSLOT = 21
};

enum Flag { F_QC = 1, F_QI = 2, F_QR = 4, F_QS = 8, F_QG = 16 };

// Warm-rain instantiation: the frozen species never leave block B / block R, so their slots (and the ones only
// the frozen-species blocks read) are not staged at all and the rest is packed into 13 physical slots:
// 12 480 B per column, 3 workgroups per CU instead of 2.  Indexed by the enumerator's VALUE; -1 = not staged.
// S0: TEMP QV RHO RC RR NR SSATW PRES NWFA NIFA NCRAW NRRAW;  S1/S2: the tendencies, RHO2, RR2, NR2.
constexpr int NSLOT_W = 13;
constexpr int WMAP[NSLOT] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, -1, 4, -1, 10, -1, 6, 11, 12, 7, -1, 8};
template <bool WARM, int S>
__device__ constexpr int slot_of()
{
    static_assert(!WARM || WMAP[S] >= 0, "this slot is not staged in the warm-rain layout");
    return WARM ? WMAP[S] : S;
}

}  // namespace

// The ~3 KB of scalar constants live in the constant address space, one slot per
// context: loads from it are invariant, so the compiler keeps them on the scalar
// unit (s_load, no vmcnt waits) instead of issuing a vector load + full drain per use.
__constant__ Consts g_consts[MAX_CONST_SLOTS];

// A global-memory pointer whose value is the same in every lane, made explicit (readfirstlane)
// so that p[lane offset] uses the "SGPR base + 32-bit VGPR offset" addressing form instead of a
// per-lane 64-bit address held in VGPRs.  The result stays in the global address space.
typedef __attribute__((address_space(1))) double gdouble;
__device__ inline gdouble *uniform_ptr(const double *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
    return (gdouble *)((static_cast<unsigned long long>(hi) << 32) | lo);
}

__device__ inline gdouble *gptr(const double *p, int64_t off)
{
    return (gdouble *)reinterpret_cast<unsigned long long>(p) + off;
}

// The kernel arguments, read where they are used.  The by-value StepArgs parameter sits at offset 0 of the
// kernarg segment; fetching a field through this (opaque) pointer is one scalar load at the point of use.
// Loading all 27 pointers at kernel entry instead keeps ~54 SGPRs live for the whole kernel, which the
// register allocator can only honour by parking them in VGPR lanes (a v_readlane pair per later use).
typedef const __attribute__((address_space(4))) StepArgs CArgs;
__device__ inline CArgs *kargs()
{
    CArgs *p = (CArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// NJ = level groups per lane (ceil(nz/64)); NL = LDS stride per slot (>= nz), a compile-time
// constant so that every slot address is "one VGPR (8k) + immediate offset (slot*NL*8)".
// CPW = columns (= waves) per workgroup.  Each wave owns one column for the passes with a vertical dependency
// (0: block E scan, 3: fall-speed scans, 4: sedimentation, 5).  The pointwise pass 1 (blocks D-N, ~60 % of the
// instructions) is shared out differently: the CPW columns of the workgroup are cut into bands of 64/CPW levels
// and one wave pass handles the same band of all CPW columns.  Lanes of a wave then sit at the same altitude,
// i.e. in the same microphysical regime (warm rain / melting layer / mixed phase / ice), so far fewer lanes idle
// in the regime-specific branches than when a wave spans 64 consecutive levels of one column.
// WARM = the context was initialised with iiwarm (namelists, M:22): the frozen-species blocks are compiled out,
// so the warm-rain kernel carries neither their code nor their registers.
template <int NJ, int NL, int CPW, bool RATES, bool WARM>
__global__ __launch_bounds__(CPW *WAVE, WARM ? 3 : 2) void thompson_column_step(const StepArgs a)
{
    constexpr int BL = WAVE / CPW;                   // levels per band
    __shared__ double Lsh[CPW][(WARM ? NSLOT_W : NSLOT) * NL];   // one [slot][NL] image per column
    __shared__ int s_alive[CPW];                     // column takes part in pass 1 (exists and has microphysics)
    __shared__ int s_next;                           // pass 1: next band to hand out
    // the cloud-droplet gamma constants indexed by nu_c (1..15, per level): LDS copies, a per-lane index into the
    // constant address space would be a vector load with a full memory round trip at each use
    __shared__ double s_cc[6][16];                   // rows: ccg(1,:), ccg(2,:), ocg1, ocg2, cce(2,:), dcg_fac
    const int wv = CPW > 1 ? __builtin_amdgcn_readfirstlane(int(threadIdx.x) / WAVE) : 0;
    double *const Lw = Lsh[wv];
#define L(slot, k) Lw[slot_of<WARM, slot>() * NL + (k)]
#define LR(slot, k) Lw[(slot) * NL + (k)]            // physical slot number (run-time slots of the frozen-species code)

    const Consts &c = g_consts[a.cslot];
    const int lane = lane_id();
    const int nz = a.nz;
    __builtin_assume(nz >= 2 && nz <= 4 * WAVE);          // checked by launch_column_step
    const int kte = nz - 1;
    const unsigned nzu = unsigned(nz), kteu = unsigned(kte);
    constexpr bool iiwarm = WARM;
    const double DT = a.dt;
    const double odt = 1. / DT, odts = 1. / DT;               // M:1277-1279 (dtsave = dt)
    const double Nt_c = c.Nt_c;
    for (int t = threadIdx.x; t < 96; t += CPW * WAVE) {  // read after the first barrier (CPW > 1) / by the same wave
        const int row = t >> 4, i = t & 15, ii = i < 15 ? i : 14;
        s_cc[row][i] = row == 0 ? c.ccg[0][ii] : row == 1 ? c.ccg[1][ii] : row == 2 ? c.ocg1[ii]
                     : row == 3 ? c.ocg2[ii] : row == 4 ? c.cce[1][ii] : c.dcg_fac[ii];
    }

    // One column per wave, one launch covers all columns (grid = ncol / CPW workgroups).  Deliberately not a
    // grid-stride loop: a loop invites the compiler to hoist every column-invariant scalar (dt-derived values,
    // the debug switches, lane predicates) to kernel entry and keep them in SGPRs across the whole body,
    // and those are the SGPRs it then spills.
    const int64_t col0 = int64_t(blockIdx.x) * CPW;          // first column of the workgroup
    const int64_t col = col0 + wv;
    bool alive = col < a.ncol;                               // wave-uniform; false: only helps out in pass 1
    {
        if (a.debug_stop == 9) return;                       // profiling aid: launch floor
        const int64_t base = col * int64_t(nz), base0 = col0 * int64_t(nz);
        // the 12 state profiles are read and written in place (no __restrict__); bases are wave-uniform and
        // rebuilt from the kernel arguments at each use (see kargs())
        // ============ pass 0: blocks B + C, M:1387-1533 ============
        int pst[NJ];              // per level group: bits 0-4 L_q* of block B, bits 8-12 L_q* of block K, bit 16 T >= 270.65
        double mvdB[NJ], rgB[NJ];
        bool any_micro = false;
        bool col_frozen = true;   // warm-rain kernel: false if the column's frozen species are all exactly zero
#pragma unroll
        for (int j = 0; j < NJ; ++j) { pst[j] = 0; mvdB[j] = 0.; rgB[j] = R1; }
        if (alive) {
        // all first-touch HBM loads of the column are issued together (one round trip):
        double i_t[NJ], i_qv[NJ], i_p[NJ], i_qc[NJ], i_qi[NJ], i_qr[NJ], i_qs[NJ], i_qg[NJ], i_ni[NJ], i_nr[NJ];
        double i_nc[NJ], i_nwfa[NJ], i_nifa[NJ];
        {
        CArgs *ka = kargs();
        const gdouble *gt = gptr(ka->t, base), *gqv = gptr(ka->qv, base), *gp = gptr(ka->p, base),
                      *gqc = gptr(ka->qc, base), *gqi = gptr(ka->qi, base), *gqr = gptr(ka->qr, base),
                      *gqs = gptr(ka->qs, base), *gqg = gptr(ka->qg, base), *gni = gptr(ka->ni, base),
                      *gnr = gptr(ka->nr, base), *gnc = gptr(ka->nc, base), *gnwfa = gptr(ka->nwfa, base),
                      *gnifa = gptr(ka->nifa, base);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
            const unsigned kc = k < nzu ? k : kteu;                 // clamped: loads stay unconditional
            i_t[j] = gt[kc];   i_qv[j] = gqv[kc]; i_p[j] = gp[kc];   i_qc[j] = gqc[kc]; i_qi[j] = gqi[kc];
            i_qr[j] = gqr[kc]; i_qs[j] = gqs[kc]; i_qg[j] = gqg[kc]; i_ni[j] = gni[kc]; i_nr[j] = gnr[kc];
            i_nc[j] = gnc[kc]; i_nwfa[j] = gnwfa[kc]; i_nifa[j] = gnifa[kc];
        }
        }
        // Warm-rain kernel: a column whose frozen species are exactly zero on input (the normal KiD warm case,
        // W:46-52) keeps them exactly zero, so pass 1 need not read them again and pass 5 need neither read nor
        // write them (bit-identical result, a third less memory traffic).
        if constexpr (iiwarm) {
            bool fz = false;
#pragma unroll
            for (int j = 0; j < NJ; ++j) fz = fz || i_qi[j] != 0. || i_qs[j] != 0. || i_qg[j] != 0. || i_ni[j] != 0.;
            col_frozen = __any(fz);                          // lanes beyond nz hold copies of the top level
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
            pst[j] = 0; mvdB[j] = 0.; rgB[j] = R1;
            if (k >= nzu) continue;
            const double temp = i_t[j];
            const double qv = fmax(1.E-10, i_qv[j]);
            const double pres = i_p[j];
            const double rho = 0.622 * pres / (Rgas * temp * (qv + 0.622));
            const double qc1 = i_qc[j], qi1 = i_qi[j], qr1 = i_qr[j], qs1 = i_qs[j], qg1 = i_qg[j];
            int f = 0;
            double ri = R1, ni = R2, rr = R1, nr = R2, rg = R1;    // rc, rs: only their flags are needed here

            if (qc1 > R1) {                                  // M:1395-1418 (nc forced to Nt_c, M:1410)
                f |= F_QC;
            }
            if (qi1 > R1) {                                  // M:1420-1445
                f |= F_QI;
                ri = qi1 * rho;
                ni = fmax(R2, i_ni[j] * rho);
                if (ni <= R2) {
                    const double lami = kc::cie[1] / 25.E-6;
                    ni = fmin(499.e3, kc::cig[0] * kc::oig2 * ri / am_i * cube(lami));
                }
                double lami = root3(am_i * kc::cig[1] * kc::oig1 * ni / ri);
                const double xDi = (bm_i + mu_i + 1.) * (1. / lami);
                if (xDi < 5.E-6) {
                    lami = kc::cie[1] / 5.E-6;
                    ni = fmin(499.e3, kc::cig[0] * kc::oig2 * ri / am_i * cube(lami));
                } else if (xDi > 300.E-6) {
                    lami = kc::cie[1] / 300.E-6;
                    ni = kc::cig[0] * kc::oig2 * ri / am_i * cube(lami);
                }
            }
            if (qr1 > R1) {                                  // M:1447-1474
                f |= F_QR;
                rr = qr1 * rho;
                nr = fmax(R2, i_nr[j] * rho);
                if (nr <= R2) nr = nr_from_mvd(c, rr, 1.0E-3);
                const double lamr = root3(am_r * kc::crg[2] * kc::org2 * nr / rr);
                double mvd = (3.0 + mu_r + 0.672) / lamr;
                if (mvd > 2.5E-3) {
                    mvd = 2.5E-3;
                    nr = nr_from_mvd(c, rr, mvd);
                } else if (mvd < D0r * 0.75) {
                    mvd = D0r * 0.75;
                    nr = nr_from_mvd(c, rr, mvd);
                }
                mvdB[j] = mvd;
            }
            if (qs1 > R1) f |= F_QS;                         // M:1475-1483
            if (qg1 > R1) { f |= F_QG; rg = qg1 * rho; }     // M:1484-1492
            rgB[j] = rg;

            const double tempc = temp - 273.15;              // M:1504-1521
            const double qvs = rslf(pres, temp);
            const double qvsi = tempc <= 0.0 ? rsif(pres, temp) : qvs;
            double ssatw = qv / qvs - 1.;
            double ssati = qv / qvsi - 1.;
            if (fabs(ssatw) < eps) ssatw = 0.0;
            if (fabs(ssati) < eps) ssati = 0.0;
            if (f != 0 || ssati > 0.0) any_micro = true;
            pst[j] = f | (temp >= 270.65 ? 1 << 16 : 0);

            L(V_TEMP, k) = temp;  L(V_QV, k) = i_qv[j];   L(V_RHO, k) = rho;     // qv raw: block K needs qv1d itself
            // the cleaned mixing ratios (block B zeroes q <= R1, M:1412...) go to LDS; pass 1 rebuilds
            // rc..rg = q*rho from them with the same product, and block J reads them directly
            L(V_RC, k) = (f & F_QC) ? qc1 : 0.;  L(V_RR, k) = (f & F_QR) ? qr1 : 0.;
            L(V_NR, k) = nr;      L(V_SSATW, k) = ssatw;
            L(V_PRES, k) = pres;  L(V_NWFA, k) = i_nwfa[j]; L(V_NIFA, k) = i_nifa[j];
            L(V_NCRAW, k) = i_nc[j]; L(V_NRRAW, k) = i_nr[j];
            if constexpr (!iiwarm) {
                L(V_RI, k) = (f & F_QI) ? qi1 : 0.;  L(V_RS, k) = (f & F_QS) ? qs1 : 0.;  L(V_RG, k) = (f & F_QG) ? qg1 : 0.;
                L(V_NI, k) = ni;      L(V_QVSI, k) = qvsi;  L(V_SSATI, k) = ssati;  L(V_NIRAW, k) = i_ni[j];
                L(V_DIFFU, k) = diffusivity(temp, pres);     // M:1522 (only the frozen-species block reads it)
            }
        }

        // ---- no_micro early return, M:1540.  Block B has already zeroed the
        //      species at or below R1 in the caller's arrays (M:1412-1413 ...).
        if (!__any(any_micro)) {
            CArgs *ka = kargs();
            gdouble *gqc = gptr(ka->qc, base), *gnc = gptr(ka->nc, base), *gqi = gptr(ka->qi, base),
                    *gni = gptr(ka->ni, base), *gqr = gptr(ka->qr, base), *gnr = gptr(ka->nr, base),
                    *gqs = gptr(ka->qs, base), *gqg = gptr(ka->qg, base);
            gdouble *grates = RATES ? gptr(ka->rates, col * int64_t(KIDMP_NRATES_) * nz) : nullptr;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                if (k >= nzu) continue;
                gqc[k] = 0.0; gnc[k] = 0.0;
                gqr[k] = 0.0; gnr[k] = 0.0;
                if (col_frozen) { gqi[k] = 0.0; gni[k] = 0.0; gqs[k] = 0.0; gqg[k] = 0.0; }
                if (RATES)
                    for (int r = 0; r < KIDMP_NRATES_; ++r) grates[int64_t(r) * nz + k] = 0.;
            }
            if (ka->nstep && lane < 4) ka->nstep[col * 4 + lane] = 0;
            alive = false;
        }

        // ---- block E scan, M:1633-1649 ----
        if constexpr (!iiwarm) if (alive) {
            int k0l = 0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (pst[j] >> 16) k0l = lane + WAVE * j;
            const int k_0 = wave_max_i(k0l);
            double n0[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                n0[j] = __builtin_inf();
                if (k < nzu)
                    n0[j] = graupel_N0(int(k) > k_0 && (pst[j] & F_QR) && mvdB[j] > 100.E-6, mvdB[j], rgB[j]);
            }
            suffix_min<NJ>(n0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                if (k < nzu) L(V_N0X, k) = n0[j];
            }
        }

        }   // alive: pass 0
        if (a.debug_stop == 1) { if (alive && lane == 0) kargs()->ppt[col * 4] += LR(0, 0) + LR(12, 1); return; }   // profiling aid only
        if (CPW > 1) {
            if (lane == 0) s_alive[wv] = alive ? (col_frozen ? 3 : 1) : 0;     // bit 1: frozen species present
            if (threadIdx.x == 0) s_next = 0;
            __syncthreads();                                 // S0 images of all CPW columns complete
        }
        // ============ pass 1: blocks D-J (M:1545-2569), then K, L(snow), rain PSD, M, N (M:2574-2960) ============
        // Blocks K-N are pointwise in k and only consume the tendencies of block J, so they run in the same
        // sweep over the level: the tendencies stay in registers and no input is read twice.
        if (CPW == 1 && !alive) return;
#undef L
#define L(slot, k) Lp[slot_of<WARM, slot>() * NL + (k)]
        const int nband = (nz + BL - 1) / BL;
#pragma unroll 1
        for (int it = 0;; ++it) {
            // one wave pass = one band of every column of the workgroup; lane -> (column cw, level k).
            // Bands are handed out dynamically (their cost differs by regime), from the middle of the column
            // outwards: the mixed-phase levels are the expensive ones and should not be left for last.
            int seq = it;
            if (CPW > 1) {
                int t = 0;
                if (lane == 0) t = atomicAdd(&s_next, 1);
                seq = __builtin_amdgcn_readfirstlane(t);
            }
            if (seq >= nband) break;
            const int band = CPW > 1 ? (nband - 1) / 2 + ((seq & 1) ? (seq + 1) / 2 : -(seq / 2)) : seq;
            const int cw = CPW > 1 ? lane / BL : 0;
            const unsigned k = unsigned(band * BL) + unsigned(lane % BL);
            // No `continue` in this loop: lanes that skipped ahead to the next iteration on their own would
            // take the next band without the rest of the wave (the compiler is free to split a loop with several
            // back edges into nested loops); the wave barrier at the end pins the single join point.
            if (k < nzu && (CPW == 1 || s_alive[cw])) {
            double *const Lp = Lsh[cw];
            const unsigned gk = unsigned(cw) * nzu + k;      // level index within the workgroup's block of columns
            // block B's flags: pass 0 left the cleaned mixing ratios (0 where q <= R1) in LDS
            CArgs *ka1 = kargs();
            // cleaned frozen-species inputs: from the LDS image, or (warm layout: not staged) again from memory
            double qi1c, qs1c, qg1c, ni_b = R2, ni1_raw, qvsi = 0., ssati = 0., diffu = 0.;
            const bool frozen_here = !iiwarm || (CPW == 1 ? col_frozen : (s_alive[cw] & 2) != 0);   // this lane's column
            if constexpr (iiwarm) {
                qi1c = 0.; qs1c = 0.; qg1c = 0.; ni1_raw = 0.;
                if (frozen_here) {
                    const double rqi = gptr(ka1->qi, base0)[gk], rqs = gptr(ka1->qs, base0)[gk], rqg = gptr(ka1->qg, base0)[gk];
                    ni1_raw = gptr(ka1->ni, base0)[gk];
                    qi1c = rqi > R1 ? rqi : 0.;  qs1c = rqs > R1 ? rqs : 0.;  qg1c = rqg > R1 ? rqg : 0.;
                }
            } else {
                qi1c = L(V_RI, k);  qs1c = L(V_RS, k);  qg1c = L(V_RG, k);
                ni_b = L(V_NI, k);  ni1_raw = L(V_NIRAW, k);
                qvsi = L(V_QVSI, k);  ssati = L(V_SSATI, k);  diffu = L(V_DIFFU, k);
            }
            const bool L_qc = L(V_RC, k) > 0., L_qi = qi1c > 0., L_qr = L(V_RR, k) > 0., L_qs = qs1c > 0., L_qg = qg1c > 0.;
            const double pres = L(V_PRES, k);
            gdouble *grates = RATES ? gptr(ka1->rates, col0 * int64_t(KIDMP_NRATES_) * nz) + int64_t(cw) * KIDMP_NRATES_ * nz : nullptr;
            const double temp = L(V_TEMP, k), qv_raw = L(V_QV, k), rho = L(V_RHO, k);
            const double qv = fmax(1.E-10, qv_raw);
            const double rc = L_qc ? L(V_RC, k) * rho : R1, ri = L_qi ? qi1c * rho : R1,
                         rr = L_qr ? L(V_RR, k) * rho : R1;
            const double rs = L_qs ? qs1c * rho : R1, rg = L_qg ? qg1c * rho : R1;
            const double ni = ni_b, nr = L(V_NR, k);
            const double ssatw = L(V_SSATW, k);
            const double nc = L_qc ? Nt_c : 2.;

            // cheap thermodynamics of block C recomputed here, M:1504-1532
            const double tempc = temp - 273.15;
            const double rhof = fm::sqrt_pos(rho_not / rho);
            const double rhof2 = fm::sqrt_pos(rhof);
            const double visco = visc_air(tempc);
            const double vsc2 = fm::sqrt_pos(rho / visco);
            const double tcond = (5.69 + 0.0168 * tempc) * 1.0E-5 * 418.936;

            // ---- D: snow moments, M:1546-1627 ----
            double smob = 0., smo0 = 0., smo1 = 0., smoc = 0., smoe = 0., smof = 0.;
            if constexpr (!iiwarm) if (L_qs) {
                const double tc0 = fmin(-0.1, temp - 273.15);
                smob = rs * kc::oams;
                const Log2Parts lsmo2 = log2_parts(smob);    // smo2 = smob since bm_s == 2 (M:1553-1554)
                {   // 0th moment, M:1571-1574
                    const double la = kc::sa[0] + kc::sa[1] * tc0 + kc::sa[4] * tc0 * tc0 + kc::sa[8] * tc0 * tc0 * tc0;
                    const double b_ = kc::sb[0] + kc::sb[1] * tc0 + kc::sb[4] * tc0 * tc0 + kc::sb[8] * tc0 * tc0 * tc0;
                    smo0 = pow10_times_pow(la, lsmo2, b_);
                }
                {   // 1st moment, M:1577-1587
                    const double la = kc::sa[0] + kc::sa[1] * tc0 + kc::sa[2] + kc::sa[3] * tc0 + kc::sa[4] * tc0 * tc0 + kc::sa[5]
                                    + kc::sa[6] * tc0 * tc0 + kc::sa[7] * tc0 + kc::sa[8] * tc0 * tc0 * tc0 + kc::sa[9];
                    const double b_ = kc::sb[0] + kc::sb[1] * tc0 + kc::sb[2] + kc::sb[3] * tc0 + kc::sb[4] * tc0 * tc0 + kc::sb[5]
                                    + kc::sb[6] * tc0 * tc0 + kc::sb[7] * tc0 + kc::sb[8] * tc0 * tc0 * tc0 + kc::sb[9];
                    smo1 = pow10_times_pow(la, lsmo2, b_);
                }
                smoc = snow_moment(lsmo2, tc0, kc::cse[0]);  // M:1590-1600
                smoe = snow_moment(lsmo2, tc0, kc::cse[12]); // M:1603-1613
                smof = snow_moment(lsmo2, tc0, kc::cse[15]); // M:1616-1626
            }

            // ---- E (per level part): graupel slope/intercept, M:1650-1653 ----
            double ilamg = 0., N0_g = 0., ig_bv = 0., ig11 = 0.;
            if constexpr (!iiwarm) {
                const double N0_exp = L(V_N0X, k);
                const double lam_exp = root4(N0_exp * am_g * kc::cgg[0] / rg);
                const double lamg = lam_exp * kc::lamg_fac;
                ilamg = 1. / lamg;
                N0_g = N0_exp / (kc::cgg[1] * lam_exp) * lamg;          // lamg**cge(2), cge(2) = 1
                if (L_qg) {
                    // ilamg**bv_g is the one general power; cge(9) = 3 + bv_g, cge(10) = 2, cge(11) = 2.5 + bv_g/2
                    ig_bv = fpow(ilamg, bv_g);
                    ig11 = ilamg * ilamg * fm::sqrt_pos(ilamg * ig_bv);
                }
            }

            // ---- F: rain slope/intercept, M:1661-1666 ----
            const double lamr0 = root3(am_r * kc::crg[2] * kc::org2 * nr / rr);
            const double ilamr = 1. / lamr0;
            double mvd_r = (3.0 + mu_r + 0.672) / lamr0;
            const double N0_r = nr * kc::org2 * lamr0;                // lamr**cre(2), cre(2) = 1
            const double lamr = 1. / ilamr;                  // "lamr = 1./ilamr(k)", M:1716 ...

            // all process rates start at zero, M:1282-1363
            double pnc_wau = 0, pnc_rcw = 0, pnc_scw = 0, pnc_gcw = 0;
            double prr_wau = 0, prr_rcw = 0, prr_rcs = 0, prr_rcg = 0, prr_sml = 0, prr_gml = 0, prr_rci = 0;
            double pnr_wau = 0, pnr_rcs = 0, pnr_rcg = 0, pnr_rci = 0, pnr_sml = 0, pnr_gml = 0, pnr_rcr = 0, pnr_rfz = 0;
            double pri_inu = 0, pni_inu = 0, pri_ihm = 0, pni_ihm = 0, pri_wfz = 0, pni_wfz = 0, pri_rfz = 0,
                   pni_rfz = 0, pri_ide = 0, pni_ide = 0, pri_rci = 0, pni_rci = 0, pni_sci = 0, pni_iau = 0;
            const double pri_iha = 0, pni_iha = 0;           // Koop freezing needs is_aerosol_aware (M:2105)
            double prs_iau = 0, prs_sci = 0, prs_rcs = 0, prs_scw = 0, prs_sde = 0, prs_ihm = 0, prs_ide = 0;
            double prg_scw = 0, prg_rfz = 0, prg_gde = 0, prg_gcw = 0, prg_rci = 0, prg_rcs = 0, prg_rcg = 0, prg_ihm = 0;
            double vts_boost = 1.5;                          // M:1751 (only read when .not.iiwarm)
            // Number rates are final when computed (no limiter touches them except pni_ide), so they are
            // folded into the three number tendencies at once instead of staying live until block J:
            //   ncten = -(nc_m)*orho, niten = (ni_p + pni_ide - ni_m)*orho, nrten = (nr_p - nr_m)*orho.
            // The terms are those of M:2422-2423, M:2457-2459, M:2508-2510; only the order of the additions
            // differs (an ulp-level change), which frees ~30 VGPRs in the hottest part of the kernel.
            double nc_m = 0., ni_p = 0., ni_m = 0., nr_p = 0., nr_m = 0.;

            // ---- G: warm-rain terms, M:1676-1726 ----
            if (L_qr && mvd_r > D0r) {
                const double Ef_rr = 1.0 - fm::exp(2300.0 * (mvd_r - 1950.0E-6));
                pnr_rcr = Ef_rr * 2.0 * nr * rr;
                nr_m += pnr_rcr;
            }
            double mvd_c = D0c;
            int nu_c = 15;
            double lamc = 1., xDc = 0.;
            if (L_qc) {
                nu_c = int(lround(1000.E6 / nc)) + 2;
                nu_c = nu_c < 15 ? nu_c : 15;
                xDc = fmax(D0c * 1.E6, root3(rc / (am_r * nc)) * 1.E6);
                lamc = root3(nc * am_r * s_cc[1][nu_c - 1] * s_cc[2][nu_c - 1] / rc);
                mvd_c = (3.0 + nu_c + 0.672) / lamc;
            }
            if (rc > 0.01e-3) {                              // Berry & Reinhardt, M:1698-1712
                const double Dc_g = (s_cc[5][nu_c - 1] / lamc) * 1.E6;
                const double Dc_b = root6(xDc * xDc * xDc * Dc_g * Dc_g * Dc_g - xDc * xDc * xDc * xDc * xDc * xDc);
                const double zq = 6.25E-6 * xDc * Dc_b * Dc_b * Dc_b - 0.4;
                const double zeta1 = 0.5 * (zq + fabs(zq));
                const double zeta = 0.027 * rc * zeta1;
                const double tq = 0.5 * Dc_b - 7.5;
                const double taud = 0.5 * (tq + fabs(tq)) + R1;
                const double tau = 3.72 / (rc * taud);
                prr_wau = zeta / tau;
                prr_wau = fmin(rc * odts, prr_wau);
                pnr_wau = prr_wau / (am_r * nu_c * D0r * D0r * D0r);
                pnc_wau = fmin(nc * odts, prr_wau / (am_r * mvd_c * mvd_c * mvd_c));
                nr_p += pnr_wau;
                nc_m += pnc_wau;
            }
            if (L_qr && mvd_r > D0r && mvd_c > D0c) {        // accretion, M:1715-1726
                int idx = 1 + int(nbins * fm::log(mvd_r / kc::Dr1) / kc::log_Drn_Dr1);
                idx = idx < nbins ? idx : nbins;
                idx = idx > 1 ? idx : 1;
                int jc = int(mvd_c * 1.E6);
                jc = jc < 1 ? 1 : (jc > nbins ? nbins : jc);
                const double Ef_rw = kargs()->tables.t_Efrw[(idx - 1) + nbins * (jc - 1)];
                const double coll = 1. / pw4(lamr + fv_r);               // (lamr+fv_r)**(-cre(9)), cre(9) = 4
                prr_rcw = rhof * kc::t1_qr_qc * Ef_rw * rc * N0_r * coll;
                prr_rcw = fmin(rc * odts, prr_rcw);
                pnc_rcw = rhof * kc::t1_qr_qc * Ef_rw * nc * N0_r * coll;
                pnc_rcw = fmin(nc * odts, pnc_rcw);
                nc_m += pnc_rcw;
            }
            // rain/snow/graupel scavenging of aerosols (M:1729-1740, M:1938-1959) only
            // feeds nwfaten/nifaten under is_aerosol_aware (M:2398-2408): not computed.

            // ---- H: frozen species, M:1749-2286 ----
            if (!iiwarm) {
                int idx_tc = int(lround(-tempc));
                idx_tc = idx_tc < 45 ? idx_tc : 45;
                idx_tc = idx_tc > 1 ? idx_tc : 1;
                int idx_t = int((tempc - 2.5) / 5.) - 1;
                idx_t = 1 > -idx_t ? 1 : -idx_t;
                idx_t = idx_t < ntb_t ? idx_t : ntb_t;

                const int idx_c = rc > kc::r_c1 ? decade_idx(rc, kc::nic2, ntb_c) : 1;
                const int idx_i = ri > kc::r_i1 ? decade_idx(ri, kc::nii2, ntb_i) : 1;
                const int idx_i1 = ni > kc::Nt_i1 ? decade_idx(ni, kc::nii3, ntb_i1) : 1;
                int idx_r = 1, idx_r1 = ntb_r1;
                if (rr > kc::r_r1) {
                    idx_r = decade_idx(rr, kc::nir2, ntb_r);
                    const double lam_exp = lamr * kc::lamr_exp_fac;
                    const double N0_exp = kc::org1 * rr / am_r * pw4(lam_exp);   // **cre(1) = 4
                    idx_r1 = decade_idx(N0_exp, kc::nir3, ntb_r1);
                }
                const int idx_s = rs > kc::r_s1 ? decade_idx(rs, kc::nis2, ntb_s) : 1;
                int idx_g = 1, idx_g1 = ntb_g1;
                if (rg > kc::r_g1) {
                    idx_g = decade_idx(rg, kc::nig2, ntb_g);
                    const double lamg = 1. / ilamg;
                    const double lam_exp = lamg * kc::lamg_exp_fac;
                    const double N0_exp = kc::ogg1 * rg / am_g * pw4(lam_exp);   // **cge(1) = 4
                    idx_g1 = decade_idx(N0_exp, kc::nig3, ntb_g1);
                }

                // Srivastava & Coen prefactor over ice, M:1884-1900
                const double otemp = 1. / temp;
                const double rvs = rho * qvsi;
                const double h1 = otemp * (lsub * otemp * oRv - 1.);
                const double rvs_p = rvs * otemp * (lsub * otemp * oRv - 1.);
                const double rvs_pp = rvs * (h1 * h1 + (-2. * lsub * otemp * otemp * otemp * oRv) + otemp * otemp);
                const double gamsc = lsub * diffu / tcond * rvs_p;
                double alphsc = 0.5 * (gamsc / (1. + gamsc)) * (gamsc / (1. + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
                alphsc = fmax(1.E-9, alphsc);
                double xsat = ssati;
                if (fabs(xsat) < 1.E-9) xsat = 0.;
                const double t1_subl = 4. * PI * (1.0 - alphsc * xsat + 2. * alphsc * alphsc * xsat * xsat
                                                  - 5. * alphsc * alphsc * alphsc * xsat * xsat * xsat) / (1. + gamsc);

                // riming, M:1903-1935
                if (L_qc && mvd_c > D0c) {
                    const double xDs = L_qs ? smoc / smob : 0.0;
                    int jc = int(mvd_c * 1.E6);
                    jc = jc < 1 ? 1 : (jc > nbins ? nbins : jc);
                    if (xDs > D0s) {
                        int idx = 1 + int(nbins * fm::log(xDs / kc::Ds1) / kc::log_Dsn_Ds1);
                        idx = idx < nbins ? idx : nbins;
                        idx = idx > 1 ? idx : 1;
                        const double Ef_sw = kargs()->tables.t_Efsw[(idx - 1) + nbins * (jc - 1)];
                        prs_scw = rhof * kc::t1_qs_qc * Ef_sw * rc * smoe;
                        pnc_scw = rhof * kc::t1_qs_qc * Ef_sw * nc * smoe;
                        pnc_scw = fmin(nc * odts, pnc_scw);
                        nc_m += pnc_scw;
                    }
                    if (rg >= kc::r_g1 && mvd_c > D0c) {
                        const double xDg = (bm_g + mu_g + 1.) * ilamg;
                        const double vtg = rhof * av_g * kc::cgg[5] * kc::ogg3 * ig_bv;
                        const double stoke_g = mvd_c * mvd_c * vtg * rho_w / (9. * visco * xDg);
                        if (xDg > D0g) {
                            double Ef_gw = 0.;
                            if (stoke_g >= 0.4 && stoke_g <= 10.) Ef_gw = 0.55 * fm::log10(2.51 * stoke_g);
                            else if (stoke_g < 0.4)               Ef_gw = 0.0;
                            else if (stoke_g > 10)                Ef_gw = 0.77;
                            const double ig9 = cube(ilamg) * ig_bv;          // ilamg**cge(9), cge(9) = 3 + bv_g
                            prg_gcw = rhof * kc::t1_qg_qc * Ef_gw * rc * N0_g * ig9;
                            pnc_gcw = rhof * kc::t1_qg_qc * Ef_gw * nc * N0_g * ig9;
                            pnc_gcw = fmin(nc * odts, pnc_gcw);
                            nc_m += pnc_gcw;
                        }
                    }
                }

                // rain <-> snow / graupel collection from the tables, M:1964-2019
                if (rr >= kc::r_r1) {
                    if (rs >= kc::r_s1) {
                        const int64_t id = (idx_s - 1) + int64_t(ntb_s) * ((idx_t - 1) + int64_t(ntb_t) * ((idx_r1 - 1) + int64_t(ntb_r1) * (idx_r - 1)));
                        const double *r = kargs()->tables.racs_rec + id * RACS_REC;
                        const double tmr_racs1 = r[0], tcr_sacr1 = r[1], tmr_racs2 = r[2], tcr_sacr2 = r[3],
                                     tcs_racs1 = r[4], tms_sacr1 = r[5];
                        if (temp < T_0) {
                            prr_rcs = -(tmr_racs2 + tcr_sacr2 + tmr_racs1 + tcr_sacr1);
                            prs_rcs = tmr_racs2 + tcr_sacr2 - tcs_racs1 - tms_sacr1;
                            prg_rcs = tmr_racs1 + tcr_sacr1 + tcs_racs1 + tms_sacr1;
                            prr_rcs = fmax(-rr * odts, prr_rcs);
                            prs_rcs = fmax(-rs * odts, prs_rcs);
                            prg_rcs = fmin((rr + rs) * odts, prg_rcs);
                            pnr_rcs = r[6] + r[7] + r[8] + r[9];
                        } else {
                            prs_rcs = -tcs_racs1 - tms_sacr1 + tmr_racs2 + tcr_sacr2;
                            prs_rcs = fmax(-rs * odts, prs_rcs);
                            prr_rcs = -prs_rcs;
                            pnr_rcs = r[7] + r[9];
                        }
                        pnr_rcs = fmin(nr * odts, pnr_rcs);
                        nr_m += pnr_rcs;
                    }
                    if (rg >= kc::r_g1) {
                        const int64_t id = (idx_g1 - 1) + int64_t(ntb_g1) * ((idx_g - 1) + int64_t(ntb_g) * ((idx_r1 - 1) + int64_t(ntb_r1) * (idx_r - 1)));
                        const double *r = kargs()->tables.racg_rec + id * RACG_REC;
                        if (temp < T_0) {
                            prg_rcg = r[0] + r[1];
                            prg_rcg = fmin(rr * odts, prg_rcg);
                            prr_rcg = -prg_rcg;
                            pnr_rcg = r[2] + r[3];
                            pnr_rcg = fmin(nr * odts, pnr_rcg);
                        } else {
                            prr_rcg = r[4];
                            prr_rcg = fmin(rg * odts, prr_rcg);
                            prg_rcg = -prr_rcg;
                            pnr_rcg = -5. * r[3];
                        }
                        nr_m += pnr_rcg;
                    }
                }

                if (temp < T_0) {                            // ---- below freezing, M:2025-2231 ----
                    vts_boost = 1.0;
                    const double rate_max = (qv - qvsi) * rho * odts * 0.999;

                    if (rr > kc::r_r1) {                       // Bigg freezing, M:2066-2086
                        const int64_t id = (idx_r - 1) + int64_t(ntb_r) * ((idx_r1 - 1) + int64_t(ntb_r1) * (idx_tc - 1));
                        const double *r = kargs()->tables.qrfz_rec + id * QRFZ_REC;
                        prg_rfz = r[0] * odts;
                        pri_rfz = r[1] * odts;
                        pni_rfz = r[2] * odts;
                        pnr_rfz = r[3] * odts;
                        pnr_rfz = fmin(nr * odts, pnr_rfz);
                        nr_m += pnr_rfz;
                        ni_p += pni_rfz;
                    } else if (rr > R1 && temp < HGFR) {
                        pri_rfz = rr * odts;
                        pnr_rfz = nr * odts;
                        pni_rfz = pnr_rfz;
                        nr_m += pnr_rfz;
                        ni_p += pni_rfz;
                    }
                    if (rc > kc::r_c1) {
                        const int id = (idx_c - 1) + ntb_c * (idx_tc - 1);
                        pri_wfz = kargs()->tables.tpi_qcfz[id] * odts;
                        pri_wfz = fmin(rc * odts, pri_wfz);
                        pni_wfz = kargs()->tables.tni_qcfz[id] * odts;
                        pni_wfz = fmin(fmin(Nt_c * odts, pri_wfz / (2. * xm0i)), pni_wfz);
                        nc_m += pni_wfz;
                        ni_p += pni_wfz;
                    } else if (rc > R1 && temp < HGFR) {
                        pri_wfz = rc * odts;
                        pni_wfz = nc * odts;
                        nc_m += pni_wfz;
                        ni_p += pni_wfz;
                    }

                    // Cooper nucleation, M:2090-2101
                    if ((ssati >= 0.25) || (ssatw > eps && temp < 253.15)) {
                        const double xnc = fmin(250.E3, TNO * fm::exp(ATO * (T_0 - temp)));
                        const double xni = ni + (pni_rfz + pni_wfz) * DT;
                        pni_inu = 0.5 * (xnc - xni + fabs(xnc - xni)) * odts;
                        pri_inu = fmin(rate_max, xm0i * pni_inu);
                        pni_inu = pri_inu / xm0i;
                        ni_p += pni_inu;
                    }

                    if (L_qi) {                              // M:2116-2149 and M:2178-2202
                        const double lami = root3(am_i * kc::cig[1] * kc::oig1 * ni / ri);
                        const double ilami = 1. / lami;
                        const double xDi = fmax(kc::D0i, (bm_i + mu_i + 1.) * ilami);
                        const double xmi = am_i * cube(xDi);
                        const double oxmi = 1. / xmi;
                        pri_ide = C_cube * t1_subl * diffu * ssati * rvs * kc::oig1 * kc::cig[4] * ni * ilami;
                        const int id = (idx_i - 1) + ntb_i * (idx_i1 - 1);
                        if (pri_ide < 0.0) {
                            pri_ide = fmax(fmax(-ri * odts, pri_ide), rate_max);
                            pni_ide = pri_ide * oxmi;
                            pni_ide = fmax(-ni * odts, pni_ide);
                        } else {
                            pri_ide = fmin(pri_ide, rate_max);
                            const double frac = kargs()->tables.tpi_ide[id];
                            prs_ide = (1.0 - frac) * pri_ide;
                            pri_ide = frac * pri_ide;
                        }
                        if ((idx_i == ntb_i) || (xDi > 5.0 * D0s)) {
                            prs_iau = ri * .99 * odts;
                            pni_iau = ni * .95 * odts;
                        } else if (xDi < 0.1 * D0s) {
                            prs_iau = 0.;
                            pni_iau = 0.;
                        } else {
                            prs_iau = kargs()->tables.tps_iaus[id] * odts;
                            prs_iau = fmin(ri * .99 * odts, prs_iau);
                            pni_iau = kargs()->tables.tni_iaus[id] * odts;
                            pni_iau = fmin(ni * .95 * odts, pni_iau);
                        }
                        ni_m += pni_iau;
                        if (rs >= kc::r_s1) {
                            prs_sci = kc::t1_qs_qi * rhof * Ef_si * ri * smoe;
                            pni_sci = prs_sci * oxmi;
                            ni_m += pni_sci;
                        }
                        if (rr >= kc::r_r1 && mvd_r > 4. * xDi) {
                            const double c9 = 1. / pw4(lamr + fv_r);
                            pri_rci = rhof * kc::t1_qr_qi * Ef_ri * ri * N0_r * c9;
                            pnr_rci = rhof * kc::t1_qr_qi * Ef_ri * ni * N0_r * c9;
                            pni_rci = pri_rci * oxmi;
                            ni_m += pni_rci;
                            nr_m += pnr_rci;
                            prr_rci = rhof * kc::t2_qr_qi * Ef_ri * ni * N0_r * (1. / pw7(lamr + fv_r));   // **(-cre(8)), cre(8) = 7
                            prr_rci = fmin(rr * odts, prr_rci);
                            prg_rci = pri_rci + prr_rci;
                        }
                    }

                    if (L_qs) {                              // snow deposition, M:2153-2164
                        double C_snow = C_sqrd + (tempc + 1.5) * (C_cube - C_sqrd) / (-30. + 1.5);
                        C_snow = fmax(C_sqrd, fmin(C_snow, C_cube));
                        prs_sde = C_snow * t1_subl * diffu * ssati * rvs * (kc::t1_qs_sd * smo1 + kc::t2_qs_sd * rhof2 * vsc2 * smof);
                        if (prs_sde < 0.) prs_sde = fmax(fmax(-rs * odts, prs_sde), rate_max);
                        else              prs_sde = fmin(prs_sde, rate_max);
                    }
                    if (L_qg && ssati < -eps) {              // graupel sublimation, M:2166-2175
                        prg_gde = C_cube * t1_subl * diffu * ssati * rvs * N0_g
                                * (kc::t1_qg_sd * (ilamg * ilamg) + kc::t2_qg_sd * vsc2 * rhof2 * ig11);
                        if (prg_gde < 0.) prg_gde = fmax(fmax(-rg * odts, prg_gde), rate_max);
                        else              prg_gde = fmin(prg_gde, rate_max);
                    }

                    if (prg_gcw > eps && tempc > -8.0) {     // Hallett-Mossop, M:2205-2218
                        double tf = 0.;
                        if (tempc >= -5.0 && tempc < -3.0)      tf = 0.5 * (-3.0 - tempc);
                        else if (tempc > -8.0 && tempc < -5.0)  tf = 0.33333333 * (8.0 + tempc);
                        pni_ihm = 3.5E8 * tf * prg_gcw;
                        pri_ihm = xm0i * pni_ihm;
                        ni_p += pni_ihm;
                        prs_ihm = prs_scw / (prs_scw + prg_gcw) * pri_ihm;
                        prg_ihm = prg_gcw / (prs_scw + prg_gcw) * pri_ihm;
                    }
                    if (prs_scw > 2.0 * prs_sde && prs_sde > eps) {   // rimed snow -> graupel, M:2224-2231
                        const double r_frac = fmin(30.0, prs_scw / prs_sde);
                        const double g_frac = fmin(0.95, 0.15 + (r_frac - 2.) * .028);
                        vts_boost = fmin(1.5, 1.1 + (r_frac - 2.) * .016);
                        prg_scw = g_frac * prs_scw;
                        prs_scw = (1. - g_frac) * prs_scw;
                    }
                } else {                                     // ---- melting, M:2237-2281 ----
                    const double delQvs = fmax(0.0, rslf(pres, 273.15) - qv);   // M:1508, only read here
                    if (L_qs) {
                        prr_sml = (tempc * tcond - lvap0 * diffu * delQvs) * (kc::t1_qs_me * smo1 + kc::t2_qs_me * rhof2 * vsc2 * smof);
                        prr_sml = prr_sml + 4218. * olfus * tempc * (prr_rcs + prs_scw);
                        prr_sml = fmin(rs * odts, fmax(0., prr_sml));
                        pnr_sml = smo0 / rs * prr_sml * fm::exp10(-0.25 * tempc);
                        pnr_sml = fmin(smo0 * odts, pnr_sml);
                        nr_p += pnr_sml;
                        if (ssati < 0.) {
                            prs_sde = C_cube * t1_subl * diffu * ssati * rvs * (kc::t1_qs_sd * smo1 + kc::t2_qs_sd * rhof2 * vsc2 * smof);
                            prs_sde = fmax(-rs * odts, prs_sde);
                        }
                    }
                    if (L_qg) {
                        const double ig10 = ilamg * ilamg;
                        prr_gml = (tempc * tcond - lvap0 * diffu * delQvs) * N0_g * (kc::t1_qg_me * ig10 + kc::t2_qg_me * rhof2 * vsc2 * ig11);
                        prr_gml = fmin(rg * odts, fmax(0., prr_gml));
                        pnr_gml = N0_g * kc::cgg[1] * ilamg / rg * prr_gml * fm::exp10(-0.5 * tempc);
                        nr_p += pnr_gml;
                        if (ssati < 0.) {
                            prg_gde = C_cube * t1_subl * diffu * ssati * rvs * N0_g * (kc::t1_qg_sd * ig10 + kc::t2_qg_sd * vsc2 * rhof2 * ig11);
                            prg_gde = fmax(-rg * odts, prg_gde);
                        }
                    }
                    if (DT > 120.) {                         // M:2277-2281
                        prr_rcw = prr_rcw + prs_scw + prg_gcw;
                        prs_scw = 0.;
                        prg_gcw = 0.;
                    }
                }
            }

            // inputs of block J: requested now, consumed after the limiters
            const double nc1_raw = L(V_NCRAW, k), nr1_raw = L(V_NRRAW, k);
            const double qc1 = L(V_RC, k), qi1 = qi1c, qr1 = L(V_RR, k);

            // ---- I: conservation limiters, M:2297-2385 ----
            {
                double sump = pri_inu + pri_ide + prs_ide + prs_sde + prg_gde + pri_iha;
                double rate_max = (qv - qvsi) * odts * 0.999;
                if ((sump > eps && sump > rate_max) || (sump < -eps && sump < rate_max)) {
                    const double ratio = rate_max / sump;
                    pri_inu *= ratio; pri_ide *= ratio; pni_ide *= ratio; prs_ide *= ratio;
                    prs_sde *= ratio; prg_gde *= ratio;
                }
                sump = -prr_wau - pri_wfz - prr_rcw - prs_scw - prg_scw - prg_gcw;
                rate_max = -rc * odts;
                if (sump < rate_max && L_qc) {
                    const double ratio = rate_max / sump;
                    prr_wau *= ratio; pri_wfz *= ratio; prr_rcw *= ratio; prs_scw *= ratio; prg_scw *= ratio; prg_gcw *= ratio;
                }
                sump = pri_ide - prs_iau - prs_sci - pri_rci;
                rate_max = -ri * odts;
                if (sump < rate_max && L_qi) {
                    const double ratio = rate_max / sump;
                    pri_ide *= ratio; prs_iau *= ratio; prs_sci *= ratio; pri_rci *= ratio;
                }
                sump = -prg_rfz - pri_rfz - prr_rci + prr_rcs + prr_rcg;
                rate_max = -rr * odts;
                if (sump < rate_max && L_qr) {
                    const double ratio = rate_max / sump;
                    prg_rfz *= ratio; pri_rfz *= ratio; prr_rci *= ratio; prr_rcs *= ratio; prr_rcg *= ratio;
                }
                sump = prs_sde - prs_ihm - prr_sml + prs_rcs;
                rate_max = -rs * odts;
                if (sump < rate_max && L_qs) {
                    const double ratio = rate_max / sump;
                    prs_sde *= ratio; prs_ihm *= ratio; prr_sml *= ratio; prs_rcs *= ratio;
                }
                sump = prg_gde - prg_ihm - prr_gml + prg_rcg;
                rate_max = -rg * odts;
                if (sump < rate_max && L_qg) {
                    const double ratio = rate_max / sump;
                    prg_gde *= ratio; prg_ihm *= ratio; prr_gml *= ratio; prg_rcg *= ratio;
                }
                pri_ihm = prs_ihm + prg_ihm;                 // M:2377-2385
                double ratio = fmin(fabs(prr_rcg), fabs(prg_rcg));
                prr_rcg = ratio * copysign(1.0, prr_rcg);
                prg_rcg = -prr_rcg;
                if (temp > T_0) {
                    ratio = fmin(fabs(prr_rcs), fabs(prs_rcs));
                    prr_rcs = ratio * copysign(1.0, prr_rcs);
                    prs_rcs = -prr_rcs;
                }
            }

            // ---- J: tendencies and number re-balancing, M:2393-2567 ----
            const double orho = 1. / rho;
            const double ocp = 1. / (Cp * (1. + 0.887 * qv));            // M:1529
            const double lvap = lvap0 + (2106.0 - 4218.0) * tempc;       // M:1531
            const double lfus2 = lsub - lvap;
            const double nc1 = L_qc ? nc1_raw : 0.0, ni1 = L_qi ? ni1_raw : 0.0, nr1 = L_qr ? nr1_raw : 0.0;   // as cleaned by block B

            const double qvten = (-pri_inu - pri_iha - pri_ide - prs_ide - prs_sde - prg_gde) * orho;
            const double qcten = (-prr_wau - pri_wfz - prr_rcw - prs_scw - prg_scw - prg_gcw) * orho;
            double ncten = (-nc_m) * orho;                   // M:2422-2424
            {
                const double xrc = fmax(R1, (qc1 + qcten * DT) * rho);
                double xnc = fmax(2., (nc1 + ncten * DT) * rho);
                if (xrc > R1) {
                    int nu = int(lround(1000.E6 / xnc)) + 2;
                    nu = nu < 15 ? nu : 15;
                    double lc = root3(xnc * am_r * s_cc[1][nu - 1] * s_cc[2][nu - 1] / rc);
                    const double xD = (bm_r + nu + 1.) / lc;
                    if (xD < D0c) {
                        lc = s_cc[4][nu - 1] / D0c;
                        xnc = s_cc[0][nu - 1] * s_cc[3][nu - 1] * xrc / am_r * cube(lc);
                        ncten = (xnc - nc1 * rho) * odts * orho;
                    } else if (xD > D0r * 2.) {
                        lc = s_cc[4][nu - 1] / (D0r * 2.);
                        xnc = s_cc[0][nu - 1] * s_cc[3][nu - 1] * xrc / am_r * cube(lc);
                        ncten = (xnc - nc1 * rho) * odts * orho;
                    }
                } else {
                    ncten = -nc1 * odts;
                }
                xnc = fmax(0., (nc1 + ncten * DT) * rho);
                if (xnc > Nt_c_max) ncten = (Nt_c_max - nc1 * rho) * odts * orho;
            }

            const double qiten = (pri_inu + pri_iha + pri_ihm + pri_wfz + pri_rfz + pri_ide - prs_iau - prs_sci - pri_rci) * orho;
            double niten = (ni_p + pni_iha + pni_ide - ni_m) * orho;          // M:2457-2460
            {
                const double xri = fmax(R1, (qi1 + qiten * DT) * rho);
                double xni = fmax(R2, (ni1 + niten * DT) * rho);
                if (xri > R1) {
                    double lami = root3(am_i * kc::cig[1] * kc::oig1 * xni / xri);
                    const double xDi = (bm_i + mu_i + 1.) * (1. / lami);
                    if (xDi < 5.E-6) {
                        lami = kc::cie[1] / 5.E-6;
                        xni = fmin(499.e3, kc::cig[0] * kc::oig2 * xri / am_i * cube(lami));
                        niten = (xni - ni1 * rho) * odts * orho;
                    } else if (xDi > 300.E-6) {
                        lami = kc::cie[1] / 300.E-6;
                        xni = kc::cig[0] * kc::oig2 * xri / am_i * cube(lami);
                        niten = (xni - ni1 * rho) * odts * orho;
                    }
                } else {
                    niten = -ni1 * odts;
                }
                xni = fmax(0., (ni1 + niten * DT) * rho);
                if (xni > 499.E3) niten = (499.E3 - ni1 * rho) * odts * orho;
            }

            double qrten = (prr_wau + prr_rcw + prr_sml + prr_gml + prr_rcs + prr_rcg - prg_rfz - pri_rfz - prr_rci) * orho;
            double nrten = (nr_p - nr_m) * orho;             // M:2508-2511
            {
                const double xrr = fmax(R1, (qr1 + qrten * DT) * rho);
                double xnr = fmax(R2, (nr1 + nrten * DT) * rho);
                if (xrr > R1) {
                    const double lr = root3(am_r * kc::crg[2] * kc::org2 * xnr / xrr);
                    mvd_r = (3.0 + mu_r + 0.672) / lr;
                    if (mvd_r > 2.5E-3) {
                        xnr = nr_from_mvd(c, xrr, 2.5E-3);
                        nrten = (xnr - nr1 * rho) * odts * orho;
                    } else if (mvd_r < D0r * 0.75) {
                        xnr = nr_from_mvd(c, xrr, D0r * 0.75);
                        nrten = (xnr - nr1 * rho) * odts * orho;
                    }
                } else {
                    qrten = -qr1 * odts;
                    nrten = -nr1 * odts;
                }
            }

            const double qsten = (prs_iau + prs_sde + prs_sci + prs_scw + prs_rcs + prs_ide - prs_ihm - prr_sml) * orho;
            const double qgten = (prg_scw + prg_rfz + prg_gde + prg_rcg + prg_gcw + prg_rci + prg_rcs - prg_ihm - prr_gml) * orho;
            double tten;
            if (temp < T_0) {
                tten = (lsub * ocp * (pri_inu + pri_ide + prs_ide + prs_sde + prg_gde + pri_iha)
                      + lfus2 * ocp * (pri_wfz + pri_rfz + prg_rfz + prs_scw + prg_scw + prg_gcw + prg_rcs + prs_rcs + prr_rci + prg_rcg)) * orho;
            } else {
                tten = (lfus * ocp * (-prr_sml - prr_gml - prr_rcg - prr_rcs) + lsub * ocp * (prs_sde + prg_gde)) * orho;
            }

            if (RATES) {                                     // save_dg order of M:2967-3119 (two of them in pass 2)
                gdouble *g = grates + k;
                if (!iiwarm) {
                    const double v[30] = {pri_inu, pri_ide, prs_ide, prs_sde, prg_gde, pri_wfz, prs_scw, prg_scw, prg_gcw, pri_ihm,
                                          pri_rfz, prs_iau, prs_sci, pri_rci, pni_inu, pni_ihm, pni_wfz, pni_rfz, pni_ide, pni_iau,
                                          pni_sci, pni_rci, prr_sml, prr_gml, pnr_rcs, pnr_rcg, pnr_rci, pnr_sml, pnr_gml, pnr_rfz};
#pragma unroll
                    for (int r = 0; r < 30; ++r) g[int64_t(r) * nz] = v[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 30; ++r) g[int64_t(r) * nz] = 0.;
                }
                g[int64_t(30) * nz] = prr_wau;
                g[int64_t(31) * nz] = prr_rcw;
                g[int64_t(33) * nz] = pnr_wau;
                g[int64_t(35) * nz] = pnr_rcr;
            }

            // ---------------- blocks K-N for the same level ----------------
            const double tten_J = tten, qvten_J = qvten, qcten_J = qcten, ncten_J = ncten, qiten_J = qiten,
                         niten_J = niten, qrten_J = qrten, nrten_J = nrten, qsten_J = qsten, qgten_J = qgten,
                         prr_gml_J = prr_gml, boost_J = vts_boost, t1_J = temp, qc1_J = qc1, qi1_J = qi1, ni1_J = ni1,
                         qr1_J = qr1, nr1_J = nr1;
            {
            const double t1 = t1_J, qv1 = qv_raw;
            const double qc1 = qc1_J, qi1 = qi1_J, ni1 = ni1_J, qr1 = qr1_J, nr1 = nr1_J, qs1 = qs1c, qg1 = qg1c;
            double tten = tten_J, qvten = qvten_J, qcten = qcten_J, ncten = ncten_J;
            const double qiten = qiten_J, niten = niten_J;
            double qrten = qrten_J, nrten = nrten_J;
            const double qsten = qsten_J, qgten = qgten_J, prr_gml = prr_gml_J;
            double nwfaten = 0.;

            // ---- K, M:2575-2655 ----
            double temp = t1 + DT * tten;
            double otemp = 1. / temp;
            double tempc = temp - 273.15;
            double qv = fmax(1.E-10, qv1 + DT * qvten);
            double rho = 0.622 * pres / (Rgas * temp * (qv + 0.622));
            double rhof = fm::sqrt_pos(rho_not / rho);
            double rhof2 = fm::sqrt_pos(rhof);
            double qvs = rslf(pres, temp);
            double ssatw = qv / qvs - 1.;
            if (fabs(ssatw) < eps) ssatw = 0.0;
            double diffu = diffusivity(temp, pres);
            double visco = visc_air(tempc);
            double vsc2 = fm::sqrt_pos(rho / visco);
            double lvap = lvap0 + (2106.0 - 4218.0) * tempc;
            double tcond = (5.69 + 0.0168 * tempc) * 1.0E-5 * 418.936;
            double ocp = 1. / (Cp * (1. + 0.887 * qv));
            const double lvt2 = lvap * lvap * ocp * oRv * otemp * otemp;

            int f2 = 0;
            double mvdK = 0.;
            double rc = R1, nc = 2., ri = R1, ni = R2, rr = R1, nr = R2, rs = R1, rg = R1;
            if ((qc1 + qcten * DT) > R1) { rc = (qc1 + qcten * DT) * rho; nc = Nt_c; f2 |= F_QC; }
            if ((qi1 + qiten * DT) > R1) {
                ri = (qi1 + qiten * DT) * rho;
                ni = fmax(R2, (ni1 + niten * DT) * rho);
                f2 |= F_QI;
            }
            if ((qr1 + qrten * DT) > R1) {
                rr = (qr1 + qrten * DT) * rho;
                nr = fmax(R2, (nr1 + nrten * DT) * rho);
                f2 |= F_QR;
                const double lr = root3(am_r * kc::crg[2] * kc::org2 * nr / rr);
                double mvd = (3.0 + mu_r + 0.672) / lr;
                if (mvd > 2.5E-3)           { mvd = 2.5E-3;      nr = nr_from_mvd(c, rr, mvd); }
                else if (mvd < D0r * 0.75)  { mvd = D0r * 0.75;  nr = nr_from_mvd(c, rr, mvd); }
                mvdK = mvd;
            }
            if ((qs1 + qsten * DT) > R1) { rs = (qs1 + qsten * DT) * rho; f2 |= F_QS; }
            if ((qg1 + qgten * DT) > R1) { rg = (qg1 + qgten * DT) * rho; f2 |= F_QG; }
            // for the second graupel-intercept scan (pass 3, M:2717-2737): rain mvd of block K (0: no rain), with
            // "temp >= 270.65" (k_0 of M:2718-2721) in the sign bit.  It crosses waves, so it goes through memory.
            if constexpr (!iiwarm) gptr(kargs()->scratch, base0)[gk] = temp >= 270.65 ? -mvdK : mvdK;

            // ---- L: snow moments needed later (smoc/smob only), M:2663-2698 ----
            double xDs = 0.;
            if constexpr (!iiwarm) if (f2 & F_QS) {
                const double tc0 = fmin(-0.1, temp - 273.15);
                const double smob = rs * kc::oams;
                const double smoc = snow_moment(log2_parts(smob), tc0, kc::cse[0]);
                xDs = smoc / smob;                           // smod (M:2701-2711) feeds nothing
            }

            // rain PSD, M:2745-2750
            const double lamrK = root3(am_r * kc::crg[2] * kc::org2 * nr / rr);
            const double ilamr = 1. / lamrK;
            const double N0_r = nr * kc::org2 * lamrK;                 // **cre(2) = 1

            // ---- M: saturation adjustment, M:2780-2873 ----
            double orho = 1. / rho;
            double prw_vcd = 0.;
            if ((ssatw > eps) || (ssatw < -eps && (f2 & F_QC))) {
                double clap = (qv - qvs) / (1. + lvt2 * qvs);
                for (int n = 0; n < 3; ++n) {
                    const double ex = fm::exp(lvt2 * clap);
                    const double fcd = qvs * ex - qv + clap;
                    const double dfcd = qvs * lvt2 * ex + 1.;
                    clap = clap - fcd / dfcd;
                }
                const double xrc = rc + clap * rho;
                double pnc_wcd = 0.;
                if (xrc > R1) {
                    prw_vcd = clap * odt;
                    if (clap > eps) {
                        const double xnc = Nt_c;
                        pnc_wcd = 0.5 * (xnc - nc + fabs(xnc - nc)) * odts * orho;
                    }
                } else {
                    prw_vcd = -rc * orho * odt;
                    pnc_wcd = -nc * orho * odt;
                }
                qvten = qvten - prw_vcd;
                qcten = qcten + prw_vcd;
                ncten = ncten + pnc_wcd;
                nwfaten = nwfaten - pnc_wcd;
                tten = tten + lvap * ocp * prw_vcd;
                rc = fmax(R1, (qc1 + DT * qcten) * rho);
                qv = fmax(1.E-10, qv1 + DT * qvten);
                temp = t1 + DT * tten;
                rho = 0.622 * pres / (Rgas * temp * (qv + 0.622));
                qvs = rslf(pres, temp);
                ssatw = qv / qvs - 1.;
            }

            // ---- N: rain evaporation, M:2880-2960 ----
            double prv_rev = 0., pnr_rev = 0.;
            if ((ssatw < -eps) && (f2 & F_QR) && (!(prw_vcd > 0.))) {
                tempc = temp - 273.15;
                otemp = 1. / temp;
                orho = 1. / rho;
                rhof = fm::sqrt_pos(rho_not * orho);
                rhof2 = fm::sqrt_pos(rhof);
                diffu = diffusivity(temp, pres);
                visco = visc_air(tempc);
                vsc2 = fm::sqrt_pos(rho / visco);
                lvap = lvap0 + (2106.0 - 4218.0) * tempc;
                tcond = (5.69 + 0.0168 * tempc) * 1.0E-5 * 418.936;
                ocp = 1. / (Cp * (1. + 0.887 * qv));

                const double rvs = rho * qvs;
                const double h1 = otemp * (lvap * otemp * oRv - 1.);
                const double rvs_p = rvs * otemp * (lvap * otemp * oRv - 1.);
                const double rvs_pp = rvs * (h1 * h1 + (-2. * lvap * otemp * otemp * otemp * oRv) + otemp * otemp);
                const double gamsc = lvap * diffu / tcond * rvs_p;
                double alphsc = 0.5 * (gamsc / (1. + gamsc)) * (gamsc / (1. + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
                alphsc = fmax(1.E-9, alphsc);
                const double xsat = fmin(-1.E-9, ssatw);
                const double t1_evap = 2. * PI * (1.0 - alphsc * xsat + 2. * alphsc * alphsc * xsat * xsat
                                                  - 5. * alphsc * alphsc * alphsc * xsat * xsat * xsat) / (1. + gamsc);
                const double lamr = 1. / ilamr;
                if (qv / qvs < 0.95 && rr * orho <= 1.E-8) {
                    prv_rev = rr * orho * odts;
                } else {
                    prv_rev = t1_evap * diffu * (-ssatw) * N0_r * rvs
                            * (kc::t1_qr_ev * (ilamr * ilamr) + kc::t2_qr_ev * vsc2 * rhof2 * (1. / cube(lamr + 0.5 * fv_r)));   // cre(10)=2, cre(11)=3
                    const double rate_max = fmin((rr * orho * odts), (qvs - qv) * odts);
                    prv_rev = fmin(rate_max, prv_rev * orho);
                    if (prr_gml > 0.0) {
                        const double eva_factor = fmin(1.0, 0.01 + (0.99 - 0.01) * (tempc / 20.0));
                        prv_rev = prv_rev * eva_factor;
                    }
                }
                pnr_rev = fmin(nr * 0.99 * orho * odts, prv_rev * nr / rr);

                qrten = qrten - prv_rev;
                qvten = qvten + prv_rev;
                nrten = nrten - pnr_rev;
                nwfaten = nwfaten + pnr_rev;
                tten = tten - lvap * ocp * prv_rev;

                rr = fmax(R1, (qr1 + DT * qrten) * rho);
                qv = fmax(1.E-10, qv1 + DT * qvten);
                nr = fmax(R2, (nr1 + DT * nrten) * rho);
                temp = t1 + DT * tten;
                rho = 0.622 * pres / (Rgas * temp * (qv + 0.622));
            }
            if (RATES) {
                grates[int64_t(32) * nz + k] = prv_rev;
                grates[int64_t(34) * nz + k] = pnr_rev;
            }

            // qv and the (inert) aerosol numbers are final here: blocks O-Q do not touch them
            CArgs *kaN = kargs();
            const double nwfa1 = L(V_NWFA, k), nifa1 = L(V_NIFA, k);
            gptr(kaN->qv, base0)[gk] = fmax(1.E-10, qv1 + qvten * DT);                                         // M:3625
            gptr(kaN->nwfa, base0)[gk] = fmax(11.1E6 / rho, fmin(9999.E6 / rho, (nwfa1 + nwfaten * DT)));   // M:3628
            gptr(kaN->nifa, base0)[gk] = fmax(naIN1 * 0.01, fmin(9999.E6 / rho, (nifa1 + 0. * DT)));       // M:3630

            L(V_TTEN, k) = tten;   L(V_QCTEN, k) = qcten; L(V_NCTEN, k) = ncten;
            L(V_QRTEN, k) = qrten; L(V_NRTEN, k) = nrten;
            if (frozen_here) {
                L(V_QITEN, k) = qiten; L(V_NITEN, k) = niten; L(V_QSTEN, k) = qsten; L(V_QGTEN, k) = qgten;
            } else {
                // warm column without frozen species: those four tendencies are exactly zero, and their slots carry
                // the cleaned inputs pass 5 adds the tendencies to, so that it does not read them from memory again
                L(V_QITEN, k) = qc1;   L(V_NITEN, k) = nc1;   L(V_QSTEN, k) = qr1;   L(V_QGTEN, k) = nr1;
            }
            if constexpr (iiwarm) L(V_QVTEN, k) = t1;        // the input temperature, likewise (slot free once qv is read)
            L(V_RHO2, k) = rho;    L(V_RR2, k) = rr;      L(V_NR2, k) = nr;
            if constexpr (!iiwarm) {
                L(V_BOOST, k) = boost_J;  L(V_TEMP2, k) = temp;  L(V_RI2, k) = ri;  L(V_NI2, k) = ni;
                L(V_RS2, k) = rs;      L(V_RG2, k) = rg;      L(V_OCP, k) = ocp;  L(V_LVAP, k) = lvap;
                // snow's mass-weighted fall speed without the riming boost, M:3290-3303: pointwise in k, so it is
                // taken here, where the lanes of a wave share an altitude, and not in the per-column fall-speed pass
                double vts0 = 0.;
                if (rs > R1) {
                    const double rhof_s = fm::sqrt_pos(rho_not / rho);
                    const double Mrat = 1. / xDs;
                    double ils1 = 1. / (Mrat * Lam0 + fv_s);
                    double ils2 = 1. / (Mrat * Lam1 + fv_s);
                    const double mm = fpow(Mrat, mu_s);
                    const double t1_vts = Kap0 * kc::csg[3] * fpow(ils1, kc::cse[3]);
                    const double t2_vts = Kap1 * mm * kc::csg[9] * fpow(ils2, kc::cse[9]);
                    ils1 = 1. / (Mrat * Lam0);
                    ils2 = 1. / (Mrat * Lam1);
                    const double t3_vts = Kap0 * kc::csg[0] * cube(ils1);          // **cse(1), cse(1) = bm_s+1 = 3
                    const double t4_vts = Kap1 * mm * kc::csg[6] * fpow(ils2, kc::cse[6]);
                    vts0 = rhof_s * av_s * (t1_vts + t2_vts) / (t3_vts + t4_vts);
                }
                L(V_VTS0, k) = vts0;
            }
            }   // blocks K-N
            }   // valid (column, level)
            __builtin_amdgcn_wave_barrier();
        }

#undef L
#define L(slot, k) Lw[slot_of<WARM, slot>() * NL + (k)]
#define LR(slot, k) Lw[(slot) * NL + (k)]            // physical slot number (run-time slots of the frozen-species code)
        if (CPW > 1) __syncthreads();                        // S1/S2 images (and the scratch profile) complete
        if (a.debug_stop == 3) { if (alive && lane == 0) kargs()->ppt[col * 4] += LR(0, 0) + LR(12, 1); return; }   // profiling aid only
        if (!alive) return;
        // Everything the remaining passes read from memory is requested here, in one round trip that the fall-speed
        // and sedimentation passes hide: dz and the block-K rain mvd (pass 3), and the column's state as block B
        // left it (pass 5 adds the tendencies to it).
        CArgs *ka = kargs();
        gdouble *gqc = gptr(ka->qc, base), *gnc = gptr(ka->nc, base), *gqi = gptr(ka->qi, base),
                *gni = gptr(ka->ni, base), *gqr = gptr(ka->qr, base), *gnr = gptr(ka->nr, base),
                *gqs = gptr(ka->qs, base), *gqg = gptr(ka->qg, base), *gt = gptr(ka->t, base);
        const gdouble *gdz = gptr(ka->dz, base), *gscr = gptr(ka->scratch, base);
        double o_qc[NJ], o_nc[NJ], o_qi[NJ], o_ni[NJ], o_qr[NJ], o_nr[NJ], o_qs[NJ], o_qg[NJ], o_t[NJ];
        double pf_dz[NJ], pf_scr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
            const unsigned kc = k < nzu ? k : kteu;
            pf_dz[j] = gdz[kc];
            pf_scr[j] = 0.;
            if constexpr (!iiwarm) pf_scr[j] = gscr[kc];     // only the graupel intercept scan reads it
            o_qc[j] = o_nc[j] = o_qr[j] = o_nr[j] = o_t[j] = 0.;
            o_qi[j] = 0.; o_ni[j] = 0.; o_qs[j] = 0.; o_qg[j] = 0.;
            if (col_frozen) {                               // (always, in the mixed-phase kernel)
                o_qc[j] = gqc[kc]; o_nc[j] = gnc[kc]; o_qr[j] = gqr[kc]; o_nr[j] = gnr[kc];
                o_qi[j] = gqi[kc]; o_ni[j] = gni[kc]; o_qs[j] = gqs[kc]; o_qg[j] = gqg[kc];
                if constexpr (!iiwarm) o_t[j] = gt[kc];
            }
        }
        // ============ pass 3: fall speeds, M:3206-3354 ============
        double vtr[NJ], vtnr[NJ], vti[NJ], vtni[NJ], vts[NJ], vtg[NJ];
        double odz[NJ], orho_[NJ], tmp2[NJ];
        int ok[NJ];
        double dzv[NJ], rhofv[NJ];   // rhof(k) = SQRT(RHO_NOT/rho(k)), refreshed once per level (M:3219)
        int nstep_r = 0, nstep_i = 0, nstep_s = 0, nstep_g = 0;
        int ksed_r = 0, ksed_i = 0, ksed_s = 0, ksed_g = 0;

#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
            vtr[j] = vtnr[j] = vti[j] = vtni[j] = vts[j] = vtg[j] = 0.;
            odz[j] = 0.; orho_[j] = 0.; tmp2[j] = 0.; ok[j] = 0; dzv[j] = 1.; rhofv[j] = 0.;
            if (k >= nzu) continue;
            const double rho = L(V_RHO2, k);
            dzv[j] = pf_dz[j];
            odz[j] = 1. / dzv[j];
            orho_[j] = 1. / rho;
            if constexpr (!iiwarm) tmp2[j] = L(V_TEMP2, k);
            const double rr = L(V_RR2, k);
            const double rhof = fm::sqrt_pos(rho_not / rho);
            rhofv[j] = rhof;
            if (rr > R1) {                                   // M:3221-3233
                ok[j] = 1;
                const double nr = L(V_NR2, k);
                const double lamr = root3(am_r * kc::crg[2] * kc::org2 * nr / rr);
                vtr[j] = rhof * av_r * kc::crg[5] * kc::org3 * pw4(lamr) * (1. / pw5(lamr + fv_r));          // cre(3)=4, cre(6)=5
                vtnr[j] = rhof * av_r * kc::crg[6] / kc::crg[11] * pw2h(lamr) * (1. / pw3h(lamr + fv_r));   // cre(12)=2.5, cre(7)=3.5
            }
        }
        carry_down<NJ, true>(vtr, vtnr, ok);
        {
            int ns = 0;
            bool falls[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                const double vm = fmax(vtr[j], vtnr[j]);
                falls[j] = k < nzu && vm > 1.E-3;            // M:3239-3243
                if (falls[j]) {
                    const double delta_tp = dzv[j] / vm;
                    const int n1 = int(DT / delta_tp + 1.);
                    ns = n1 > ns ? n1 : ns;
                }
            }
            nstep_r = wave_max_i(ns);
            ksed_r = top_level<NJ>(falls);
            if (ksed_r == kte) ksed_r = kte - 1;
        }
        double onstep_r = 1.0, onstep_i = 1.0, onstep_s = 1.0, onstep_g = 1.0;
        if (nstep_r > 0) onstep_r = 1. / double(nstep_r);    // M:3246

        double n0x2[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) n0x2[j] = gonv_max;
        if constexpr (!iiwarm) {
            // graupel slope from the second running minimum, M:2717-2737
            {
                int k0l = 0;
                double mvdK[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    const double sv = k < nzu ? pf_scr[j] : 0.;
                    mvdK[j] = fabs(sv);
                    if (k < nzu && __builtin_signbit(sv)) k0l = int(k);
                }
                const int k_0 = wave_max_i(k0l);
                double n0[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    n0[j] = __builtin_inf();
                    if (k < nzu)
                        n0[j] = graupel_N0(int(k) > k_0 && mvdK[j] > 100.E-6, mvdK[j], L(V_RG2, k));
                }
                suffix_min<NJ>(n0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) n0x2[j] = n0[j];
            }

            // ice, M:3253-3278
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                ok[j] = 0;
                if (k >= nzu) continue;
                const double ri = L(V_RI2, k);
                if (ri > R1) {
                    ok[j] = 1;
                    const double rhof = rhofv[j];
                    const double lami = root3(am_i * kc::cig[1] * kc::oig1 * L(V_NI2, k) / ri);
                    const double ilami = 1. / lami;
                    const double pw = ilami;                             // ilami**bv_i, bv_i = 1
                    vti[j] = rhof * av_i * kc::cig[2] * kc::oig2 * pw;
                    vtni[j] = rhof * av_i * kc::cig[5] / kc::cig[6] * pw;
                }
            }
            carry_down<NJ, true>(vti, vtni, ok);
            {
                int ns = 0;
                bool falls[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    falls[j] = k < nzu && vti[j] > 1.E-3;
                    if (falls[j]) {
                        const int n1 = int(DT / (dzv[j] / vti[j]) + 1.);
                        ns = n1 > ns ? n1 : ns;
                    }
                }
                nstep_i = wave_max_i(ns);
                ksed_i = top_level<NJ>(falls);
                if (ksed_i == kte) ksed_i = kte - 1;
                if (nstep_i > 0) onstep_i = 1. / double(nstep_i);
            }

            // snow, M:3284-3317
            double dummy[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                ok[j] = 0; dummy[j] = 0.;
                if (k >= nzu) continue;
                if (L(V_RS2, k) > R1) {
                    ok[j] = 1;
                    const double v = L(V_VTS0, k);           // mass-weighted fall speed before the boost, from pass 1
                    const double boost = L(V_BOOST, k);
                    if (tmp2[j] > (T_0 + 0.1))
                        vts[j] = fmax(v * boost, v * ((vtr[j] - v * boost) / (tmp2[j] - T_0)));
                    else
                        vts[j] = v * boost;
                }
            }
            carry_down<NJ, false>(vts, dummy, ok);
            {
                int ns = 0;
                bool falls[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    falls[j] = k < nzu && vts[j] > 1.E-3;
                    if (falls[j]) {
                        const int n1 = int(DT / (dzv[j] / vts[j]) + 1.);
                        ns = n1 > ns ? n1 : ns;
                    }
                }
                nstep_s = wave_max_i(ns);
                ksed_s = top_level<NJ>(falls);
                if (ksed_s == kte) ksed_s = kte - 1;
                if (nstep_s > 0) onstep_s = 1. / double(nstep_s);
            }

            // graupel, M:3321-3343
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                ok[j] = 0; dummy[j] = 0.;
                if (k >= nzu) continue;
                const double rg = L(V_RG2, k);
                if (rg > R1) {
                    ok[j] = 1;
                    const double rhof = rhofv[j];
                    const double N0_exp = n0x2[j];
                    const double lam_exp = root4(N0_exp * am_g * kc::cgg[0] / rg);
                    const double lamg = lam_exp * kc::lamg_fac;
                    const double ilamg = 1. / lamg;
                    const double v = rhof * av_g * kc::cgg[5] * kc::ogg3 * fpow(ilamg, bv_g);
                    vtg[j] = tmp2[j] > T_0 ? fmax(v, vtr[j]) : v;
                }
            }
            carry_down<NJ, false>(vtg, dummy, ok);
            {
                int ns = 0;
                bool falls[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    falls[j] = k < nzu && vtg[j] > 1.E-3;
                    if (falls[j]) {
                        const int n1 = int(DT / (dzv[j] / vtg[j]) + 1.);
                        ns = n1 > ns ? n1 : ns;
                    }
                }
                nstep_g = wave_max_i(ns);
                ksed_g = top_level<NJ>(falls);
                if (ksed_g == kte) ksed_g = kte - 1;
                if (nstep_g > 0) onstep_g = 1. / double(nstep_g);
            }
        }
        // nstep = NINT(1./onstep), M:3365 etc. (0 -> 1)
        nstep_r = int(lround(1. / onstep_r));
        nstep_i = int(lround(1. / onstep_i));
        nstep_s = int(lround(1. / onstep_s));
        nstep_g = int(lround(1. / onstep_g));
        if (int32_t *ns = kargs()->nstep; ns && lane == 0) {
            ns[col * 4 + 0] = nstep_r; ns[col * 4 + 1] = nstep_i;
            ns[col * 4 + 2] = nstep_s; ns[col * 4 + 3] = nstep_g;
        }

        if (a.debug_stop == 4) { if (lane == 0) kargs()->ppt[col * 4] += LR(0, 0) + LR(12, 1); return; }   // profiling aid only
        // ============ pass 4: sedimentation sweeps, M:3365-3578 ============
        double ppt_r = 0., ppt_s = 0., ppt_g = 0., ppt_i = 0.;
        {   // rain (never gated by l_sediment), M:3365-3399
            double r[NJ], n[NJ], qt[NJ], nt[NJ], sr[NJ], sn[NJ], ur[NJ], un[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                const bool in = k < nzu;
                r[j] = in ? L(V_RR2, k) : 0.;   n[j] = in ? L(V_NR2, k) : 0.;
                qt[j] = in ? L(V_QRTEN, k) : 0.; nt[j] = in ? L(V_NRTEN, k) : 0.;
            }
            // per-level weights of a substep: odzq*onstep*orho for the tendency, odzq*DT*onstep for the content
            // (the reference multiplies them in one by one, M:3380-3387; hoisting them differs in the last bit)
            double wten_r[NJ], wcon_r[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { wten_r[j] = odz[j] * onstep_r * orho_[j]; wcon_r[j] = odz[j] * DT * onstep_r; }
            for (int s = 0; s < nstep_r; ++s) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) { sr[j] = vtr[j] * r[j]; sn[j] = vtnr[j] * n[j]; }
                shift_from_above<NJ>(sr, ur);
                shift_from_above<NJ>(sn, un);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    // the top level (M:3370-3377) is the general update with nothing falling in: ur = un = 0 there
                    // (the slot above holds r = 0), and q - x == q + (0 - x) bit for bit
                    if (k == kteu || int(k) <= ksed_r) {
                        const double dq = ur[j] - sr[j], dn = un[j] - sn[j];
                        if constexpr (iiwarm) {              // one or two substeps: the reference's product order
                            qt[j] = qt[j] + dq * odz[j] * onstep_r * orho_[j];
                            nt[j] = nt[j] + dn * odz[j] * onstep_r * orho_[j];
                            r[j] = fmax(R1, r[j] + dq * odz[j] * DT * onstep_r);
                            n[j] = fmax(R2, n[j] + dn * odz[j] * DT * onstep_r);
                        } else {                             // many substeps: per-level weights hoisted out of the loop
                            qt[j] = qt[j] + dq * wten_r[j];
                            nt[j] = nt[j] + dn * wten_r[j];
                            r[j] = fmax(R1, r[j] + dq * wcon_r[j]);
                            n[j] = fmax(R2, n[j] + dn * wcon_r[j]);
                        }
                    }
                }
                if (r[0] > R1 * 10.) ppt_r = ppt_r + sr[0] * DT * onstep_r;   // meaningful on lane 0 (k = kts)
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                if (k < nzu) { L(V_QRTEN, k) = qt[j]; L(V_NRTEN, k) = nt[j]; }
            }
        }
        if constexpr (!iiwarm) if (c.l_sediment) {
            {   // ice, M:3447-3480
                double r[NJ], n[NJ], qt[NJ], nt[NJ], sr[NJ], sn[NJ], ur[NJ], un[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    const bool in = k < nzu;
                    r[j] = in ? L(V_RI2, k) : 0.;   n[j] = in ? L(V_NI2, k) : 0.;
                    qt[j] = in ? L(V_QITEN, k) : 0.; nt[j] = in ? L(V_NITEN, k) : 0.;
                }
                double wten_i[NJ], wcon_i[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { wten_i[j] = odz[j] * onstep_i * orho_[j]; wcon_i[j] = odz[j] * DT * onstep_i; }
                for (int s = 0; s < nstep_i; ++s) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { sr[j] = vti[j] * r[j]; sn[j] = vtni[j] * n[j]; }
                    shift_from_above<NJ>(sr, ur);
                    shift_from_above<NJ>(sn, un);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                        if (k == kteu || int(k) <= ksed_i) {     // top level: as for rain
                            const double dq = ur[j] - sr[j], dn = un[j] - sn[j];
                            qt[j] = qt[j] + dq * wten_i[j];
                            nt[j] = nt[j] + dn * wten_i[j];
                            r[j] = fmax(R1, r[j] + dq * wcon_i[j]);
                            n[j] = fmax(R2, n[j] + dn * wcon_i[j]);
                        }
                    }
                    if (r[0] > R1 * 10.) ppt_i = ppt_i + sr[0] * DT * onstep_i;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    if (k < nzu) { L(V_QITEN, k) = qt[j]; L(V_NITEN, k) = nt[j]; }
                }
            }
            // snow (M:3504-3529) and graupel (M:3553-3578): mass only
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                double r[NJ], qt[NJ], sr[NJ], ur[NJ];
                const int slot_r = sp == 0 ? V_RS2 : V_RG2, slot_t = sp == 0 ? V_QSTEN : V_QGTEN;
                const int nst = sp == 0 ? nstep_s : nstep_g, ksed = sp == 0 ? ksed_s : ksed_g;
                const double onst = sp == 0 ? onstep_s : onstep_g;
                double pp = 0.;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    const bool in = k < nzu;
                    r[j] = in ? LR(slot_r, k) : 0.;
                    qt[j] = in ? LR(slot_t, k) : 0.;
                }
                double wten[NJ], wcon[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) { wten[j] = odz[j] * onst * orho_[j]; wcon[j] = odz[j] * DT * onst; }
                for (int s = 0; s < nst; ++s) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) sr[j] = (sp == 0 ? vts[j] : vtg[j]) * r[j];
                    shift_from_above<NJ>(sr, ur);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                        if (k == kteu || int(k) <= ksed) {
                            const double dq = ur[j] - sr[j];
                            qt[j] = qt[j] + dq * wten[j];
                            r[j] = fmax(R1, r[j] + dq * wcon[j]);
                        }
                    }
                    if (r[0] > R1 * 10.) pp = pp + sr[0] * DT * onst;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
                    if (k < nzu) LR(slot_t, k) = qt[j];
                }
                if (sp == 0) ppt_s = pp; else ppt_g = pp;
            }
        }
        if (lane == 0) {                                     // precipitation is accumulated (INOUT, M:1172)
            double *pp = kargs()->ppt + col * 4;
            pp[0] = pp[0] + ppt_r;
            pp[1] = pp[1] + ppt_s;
            pp[2] = pp[2] + ppt_g;
            pp[3] = pp[3] + ppt_i;
        }

        if (a.debug_stop == 5) { if (lane == 0) kargs()->ppt[col * 4] += LR(0, 0) + LR(12, 1); return; }   // profiling aid only
        // ============ pass 5: blocks Q + R, M:3584-3686 (inputs requested before pass 3) ============
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned k = unsigned(lane) + unsigned(WAVE) * unsigned(j);
            if (k >= nzu) continue;
            const int f = pst[j] & 31;
            double rqc = o_qc[j], rnc = o_nc[j], rqr = o_qr[j], rnr = o_nr[j], t1 = o_t[j];
            const double rqi = o_qi[j], rni = o_ni[j], rqs = o_qs[j], rqg = o_qg[j];         // cleaned below (block B)
            if constexpr (iiwarm) {
                t1 = L(V_QVTEN, k);
                if (!col_frozen) { rqc = L(V_QITEN, k); rnc = L(V_NITEN, k); rqr = L(V_QSTEN, k); rnr = L(V_QGTEN, k); }
            }
            const double qc1 = (f & F_QC) ? rqc : 0.0, nc1 = (f & F_QC) ? rnc : 0.0;
            const double qi1 = (f & F_QI) ? rqi : 0.0, ni1 = (f & F_QI) ? rni : 0.0;
            const double qr1 = (f & F_QR) ? rqr : 0.0, nr1 = (f & F_QR) ? rnr : 0.0;
            const double qs1 = (f & F_QS) ? rqs : 0.0, qg1 = (f & F_QG) ? rqg : 0.0;
            double tten = L(V_TTEN, k), qcten = L(V_QCTEN, k), ncten = L(V_NCTEN, k);
            double qiten = 0., niten = 0., qsten = 0., qgten = 0.;
            if (col_frozen) { qiten = L(V_QITEN, k); niten = L(V_NITEN, k); qsten = L(V_QSTEN, k); qgten = L(V_QGTEN, k); }
            const double qrten = L(V_QRTEN, k), nrten = L(V_NRTEN, k);
            const double rho = L(V_RHO2, k);

            if constexpr (!iiwarm) {                         // Q, M:3585-3605
                const double temp = L(V_TEMP2, k), ocp = L(V_OCP, k), lvap = L(V_LVAP, k);
                const double xri = fmax(0.0, qi1 + qiten * DT);
                if ((temp > T_0) && (xri > 0.0)) {
                    qcten = qcten + xri * odt;
                    ncten = ncten + ni1 * odt;
                    qiten = qiten - xri * odt;
                    niten = -ni1 * odt;
                    tten = tten - lfus * ocp * xri * odt;
                }
                const double xrc = fmax(0.0, qc1 + qcten * DT);
                if ((temp < HGFR) && (xrc > 0.0)) {
                    const double lfus2 = lsub - lvap;
                    const double xnc = nc1 + ncten * DT;
                    qiten = qiten + xrc * odt;
                    niten = niten + xnc * odt;
                    qcten = qcten - xrc * odt;
                    ncten = ncten - xnc * odt;
                    tten = tten + lfus2 * ocp * xrc * odt;
                }
            }

            // R, M:3624-3685
            gt[k] = t1 + tten * DT;
            double qc = qc1 + qcten * DT;
            double ncn = fmax(2. / rho, nc1 + ncten * DT);
            if (qc <= R1) {
                qc = 0.0;
                ncn = 0.0;
            } else {
                int nu = int(lround(1000.E6 / (ncn * rho))) + 2;
                nu = nu < 15 ? nu : 15;
                double lc = root3(am_r * s_cc[1][nu - 1] * s_cc[2][nu - 1] * ncn / qc);
                const double xD = (bm_r + nu + 1.) / lc;
                if (xD < D0c)            lc = s_cc[4][nu - 1] / D0c;
                else if (xD > D0r * 2.)  lc = s_cc[4][nu - 1] / (D0r * 2.);
                ncn = fmin(s_cc[0][nu - 1] * s_cc[3][nu - 1] * qc / am_r * cube(lc), Nt_c_max / rho);
            }
            gqc[k] = qc;
            gnc[k] = ncn;

            double qi = qi1 + qiten * DT;
            double nin = fmax(R2 / rho, ni1 + niten * DT);
            if (qi <= R1) {
                qi = 0.0;
                nin = 0.0;
            } else {
                double lami = root3(am_i * kc::cig[1] * kc::oig1 * nin / qi);
                const double xDi = (bm_i + mu_i + 1.) * (1. / lami);
                if (xDi < 5.E-6)          lami = kc::cie[1] / 5.E-6;
                else if (xDi > 300.E-6)   lami = kc::cie[1] / 300.E-6;
                nin = fmin(kc::cig[0] * kc::oig2 * qi / am_i * cube(lami), 499.e3 / rho);
            }
            if (col_frozen) { gqi[k] = qi; gni[k] = nin; }

            double qr = qr1 + qrten * DT;
            double nrn = fmax(R2 / rho, nr1 + nrten * DT);
            if (qr <= R1) {
                qr = 0.0;
                nrn = 0.0;
            } else {
                const double lr = root3(am_r * kc::crg[2] * kc::org2 * nrn / qr);
                double mvd = (3.0 + mu_r + 0.672) / lr;
                if (mvd > 2.5E-3)           mvd = 2.5E-3;
                else if (mvd < D0r * 0.75)  mvd = D0r * 0.75;
                nrn = nr_from_mvd(c, qr, mvd);
            }
            gqr[k] = qr;
            gnr[k] = nrn;

            const double qs = qs1 + qsten * DT;
            const double qg = qg1 + qgten * DT;
            if (col_frozen) { gqs[k] = qs <= R1 ? 0.0 : qs; gqg[k] = qg <= R1 ? 0.0 : qg; }
        }
    }
}

const char *column_kernel_name() { return "thompson_column_step"; }

// The kernel uses the PARAMETER-derived constants as immediates from the generated header; check that
// the header was generated from the same host code that computed `c` (bitwise).
bool generated_consts_match(const Consts &c)
{
    auto same = [](const double *a, const double *b, int n) {
        for (int i = 0; i < n; ++i)
            if (!(a[i] == b[i])) return false;
        return true;
    };
    return same(c.cie, kc::cie, 7) && same(c.cig, kc::cig, 7) && same(c.cre, kc::cre, 13) && same(c.crg, kc::crg, 13)
        && same(c.cse, kc::cse, 18) && same(c.csg, kc::csg, 18) && same(c.cge, kc::cge, 12) && same(c.cgg, kc::cgg, 12)
        && same(c.sa, kc::sa, 10) && same(c.sb, kc::sb, 10) && c.D0i == kc::D0i && c.oig1 == kc::oig1
        && c.oig2 == kc::oig2 && c.org1 == kc::org1 && c.org2 == kc::org2 && c.org3 == kc::org3 && c.oams == kc::oams
        && c.ogg1 == kc::ogg1 && c.ogg3 == kc::ogg3 && c.t1_qr_qc == kc::t1_qr_qc && c.t2_qr_qi == kc::t2_qr_qi
        && c.t1_qg_qc == kc::t1_qg_qc && c.t2_qr_ev == kc::t2_qr_ev && c.t2_qs_sd == kc::t2_qs_sd
        && c.t2_qs_me == kc::t2_qs_me && c.t2_qg_sd == kc::t2_qg_sd && c.t2_qg_me == kc::t2_qg_me
        && c.t1_qs_me == kc::t1_qs_me && c.t1_qg_me == kc::t1_qg_me && c.t1_qg_sd == kc::t1_qg_sd
        && c.lamg_fac == kc::lamg_fac && c.lamr_exp_fac == kc::lamr_exp_fac && c.lamg_exp_fac == kc::lamg_exp_fac
        && c.Dr1 == kc::Dr1 && c.Drn == kc::Drn && c.Ds1 == kc::Ds1 && c.Dsn == kc::Dsn && c.r_c1 == kc::r_c1
        && c.r_i1 == kc::r_i1 && c.r_r1 == kc::r_r1 && c.r_s1 == kc::r_s1 && c.r_g1 == kc::r_g1 && c.Nt_i1 == kc::Nt_i1
        && c.nic2 == kc::nic2 && c.nii2 == kc::nii2 && c.nii3 == kc::nii3 && c.nir2 == kc::nir2 && c.nir3 == kc::nir3
        && c.nis2 == kc::nis2 && c.nig2 == kc::nig2 && c.nig3 == kc::nig3;
}

hipError_t upload_consts(int slot, const Consts &c)
{
    if (slot < 0 || slot >= MAX_CONST_SLOTS) return hipErrorInvalidValue;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_consts), &c, sizeof(Consts), size_t(slot) * sizeof(Consts), hipMemcpyHostToDevice);
}

template <int NJ, int NL, int CPW>
static hipError_t launch_nj(const StepArgs &a, bool rates, int grid, hipStream_t s)
{
    const int g = (grid + CPW - 1) / CPW;
    const dim3 gd(g), bd(CPW * WAVE);
    if (a.iiwarm) {
        if (rates) hipLaunchKernelGGL((thompson_column_step<NJ, NL, CPW, true, true>), gd, bd, 0, s, a);
        else       hipLaunchKernelGGL((thompson_column_step<NJ, NL, CPW, false, true>), gd, bd, 0, s, a);
    } else {
        if (rates) hipLaunchKernelGGL((thompson_column_step<NJ, NL, CPW, true, false>), gd, bd, 0, s, a);
        else       hipLaunchKernelGGL((thompson_column_step<NJ, NL, CPW, false, false>), gd, bd, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_column_step(const StepArgs &a, hipStream_t s)
{
    if (a.ncol <= 0) return hipSuccess;
    if (a.nz < 2 || a.nz > 4 * WAVE) return hipErrorInvalidValue;
    if (a.ncol > int64_t(0x7fffffff)) return hipErrorInvalidValue;   // one workgroup per column; 2^31 columns exceed HBM anyway
    const int grid = int(a.ncol);
    const bool rates = a.rates != nullptr;
    if (!a.scratch) return hipErrorInvalidValue;
    // LDS per column = 21 slots * NL * 8 B; NL = 120 (KiD's nz) gives 8 resident columns per CU, as two
    // workgroups of 4.  Taller columns run one column per workgroup (4 images would not leave two workgroups
    // per CU, or not fit at all).
    if (a.nz <= 64)  return launch_nj<1, 64, 4>(a, rates, grid, s);
    if (a.nz <= 120) {
        static const int cpw = getenv("KIDMP_CPW") ? atoi(getenv("KIDMP_CPW")) : 4;   // tuning aid: 1 = no banding
        if (cpw == 1) return launch_nj<2, 120, 1>(a, rates, grid, s);
        return launch_nj<2, 120, 4>(a, rates, grid, s);
    }
    if (a.nz <= 128) return launch_nj<2, 128, 1>(a, rates, grid, s);
    if (a.nz <= 192) return launch_nj<3, 192, 1>(a, rates, grid, s);
    return launch_nj<4, 256, 1>(a, rates, grid, s);
}

}  // namespace kidmp
