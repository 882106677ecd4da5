// fastmath.h -- fp64 log / exp / pow for the column kernel, sized to the arguments the scheme produces.
//
// The device math library's log, log10, exp, exp2 cost 60-100 fp64 instructions each because they carry
// sub-ulp extended-precision paths and every special case.  The scheme only calls them on positive, finite,
// normal arguments of moderate size, where a classic argument reduction plus one polynomial is enough:
//
//   ln m   on m in [sqrt(1/2), sqrt(2)):  s = f/(2+f), f = m-1;  ln(1+f) = f - (f^2/2 - s (f^2/2 + R(s^2)))
//          with the degree-7 minimax R of the well-known fdlibm reduction (error < 1 ulp)
//   2**r   on |r| <= 0.52: degree-13 Taylor polynomial in r (coefficients ln2^k/k!), even/odd split
//   e**r   on |r| <= 0.36: degree-13 Taylor polynomial
//
// Every function here is within ~2 ulp of the correctly rounded result on its stated domain
// (tests/test_fastmath.py measures this against 80-bit long double on the host build of this same header).
// No tables, no divergent branches.  Host + device so that the accuracy test needs no GPU.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define KFM_FN __host__ __device__ inline
#else
#define KFM_FN inline
#endif

namespace kidmp {
namespace fm {

constexpr double LN2 = 0.6931471805599453;
constexpr double LN2_HI = 0.6931471803691238, LN2_LO = 1.9082149292705877e-10;   // LN2_HI has 21 trailing zero bits
constexpr double INV_LN2 = 1.4426950408889634;
constexpr double INV_LN10 = 0.4342944819032518;
constexpr double LOG10_2 = 0.3010299956639812;
constexpr double LOG2_10_HI = 3.321928094887362, LOG2_10_LO = 1.661617516973592e-16;
constexpr double SQRT_HALF = 0.70710678118654757;

// a*b + C for a compile-time constant C, on the device as v_fma_f64 (VOP3) with C in an SGPR pair.  Left to itself the
// compiler turns a Horner step into v_fmac_f64 (VOP2, addend tied to the destination) and moves the constant into the
// destination VGPR pair first -- two v_mov_b32, VECTOR instructions, per coefficient (600 of the mixed-phase kernel's 7 500
// static VALU instructions were such moves).  Scalar moves go to the scalar unit, which has room at three waves per SIMD.
// Same operation, same rounding.  (gfx9's VOP3 reads ONE scalar operand: a step whose multiplicand is a constant too
// keeps std::fma.)
KFM_FN double fma_k(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return std::fma(a, b, c);
#endif
}

// x = 2**e * m with m in [sqrt(1/2), sqrt(2)); x positive, finite, normal
struct Split { double e, m; };
KFM_FN Split split(double x)
{
    int e;
    double m = std::frexp(x, &e);                        // m in [0.5, 1)
    const bool low = m < SQRT_HALF;
    m = low ? m * 2. : m;
    e = low ? e - 1 : e;
    return Split{double(e), m};
}

// ln(m) for m in [sqrt(1/2), sqrt(2))
KFM_FN double ln_mant(double m)
{
    constexpr double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                     Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                     Lg7 = 1.479819860511658591e-01;
    const double f = m - 1.;                             // exact
    const double s = f / (2. + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_k(w, std::fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_k(w, fma_k(w, std::fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return f - (hfsq - s * (hfsq + R));
}

// 2**r for |r| <= 0.52 (relative error < 1 ulp)
KFM_FN double exp2_small(double r)
{
    constexpr double c1 = 0.6931471805599453, c2 = 0.24022650695910072, c3 = 0.05550410866482158,
                     c4 = 0.009618129107628477, c5 = 0.0013333558146428443, c6 = 0.0001540353039338161,
                     c7 = 1.5252733804059841e-05, c8 = 1.321548679014431e-06, c9 = 1.01780860092397e-07,
                     c10 = 7.054911620801123e-09, c11 = 4.4455382718708116e-10, c12 = 2.5678435993488206e-11,
                     c13 = 1.3691488853904128e-12;
    const double q = r * r;
    // 2**r = 1 + r*(c1 + c3 q + c5 q^2 ...) + q*(c2 + c4 q + ...): two independent Horner chains in q
    const double od = fma_k(q, fma_k(q, fma_k(q, fma_k(q, fma_k(q, std::fma(q, c13, c11), c9), c7), c5), c3), c1);
    const double ev = fma_k(q, fma_k(q, fma_k(q, fma_k(q, std::fma(q, c12, c10), c8), c6), c4), c2);
    return 1. + std::fma(r, od, q * ev);
}

// e**r for |r| <= 0.36
KFM_FN double exp_small(double r)
{
    constexpr double d2 = 0.5, d3 = 0.16666666666666666, d4 = 0.041666666666666664, d5 = 0.008333333333333333,
                     d6 = 0.001388888888888889, d7 = 0.0001984126984126984, d8 = 2.48015873015873e-05,
                     d9 = 2.7557319223985893e-06, d10 = 2.755731922398589e-07, d11 = 2.505210838544172e-08,
                     d12 = 2.08767569878681e-09, d13 = 1.6059043836821613e-10;
    const double q = r * r;
    const double od = fma_k(q, fma_k(q, fma_k(q, fma_k(q, std::fma(q, d13, d11), d9), d7), d5), d3);   // r^3 and up
    const double ev = fma_k(q, fma_k(q, fma_k(q, fma_k(q, std::fma(q, d12, d10), d8), d6), d4), d2);
    return 1. + (r + q * std::fma(r, od, ev));
}

// sqrt(x) for positive, finite, normal x well inside the exponent range (no rescaling, no 0/inf cases):
// reciprocal-square-root seed, one Goldschmidt step and one residual correction (the iteration the compiler's
// own fp64 sqrt expansion uses, minus its range handling and its second correction).  <= 1 ulp.
KFM_FN double sqrt_pos(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);            // v_rsq_f64
#else
    const double y = double(1.0f / std::sqrt(float(x))); // host stand-in with a seed of comparable quality
#endif
    const double g0 = x * y, h0 = 0.5 * y;
    const double r0 = std::fma(-h0, g0, 0.5);
    const double g1 = std::fma(g0, r0, g0), h1 = std::fma(h0, r0, h0);
    const double d0 = std::fma(-g1, g1, x);
    return std::fma(d0, h1, g1);                         // g1 is good to ~2**-47 (seed 2**-24): one residual correction suffices
}

// cbrt(x) for positive x in [1e-37, 1e37] (the fp32 range: the seed is taken in fp32):
//   y0 = 2**(-log2(x)/3) from the fp32 log2/exp2 units (>= 18 bits), one cubically convergent step towards
//   x**(-1/3) (e = 1 - x y^3; y <- y (1 + e/3 + 2 e^2/9)), c = x y^2, one Newton correction of c.  <= 1 ulp.
KFM_FN double cbrt_pos(double x)
{
    const float xf = float(x);
#if defined(__HIP_DEVICE_COMPILE__)
    const float yf = __builtin_amdgcn_exp2f(-0.33333334f * __builtin_amdgcn_logf(xf));   // v_exp_f32, v_log_f32
#else
    const float yf = std::exp2(-0.33333334f * std::log2(xf));
#endif
    double y = double(yf);
    const double t0 = y * y;
    const double e = std::fma(-x, t0 * y, 1.);
    y = std::fma(y * e, std::fma(e, 2. / 9., 1. / 3.), y);
    const double t = y * y;
    const double c = x * t;
    const double r = std::fma(-(c * c), c, x);
    return std::fma(r, t * (1. / 3.), c);
}

// a / b for finite, normal b and a/b well inside the exponent range: v_rcp_f64 seed, Newton steps, one residual
// correction of the quotient (the compiler's own `afn` expansion does two Newton steps; KFM_DIV_NR selects).
#ifndef KFM_DIV_NR
#define KFM_DIV_NR 1
#endif
KFM_FN double div(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(b);
#else
    double y = double(1.0f / float(b));
#endif
    y = std::fma(std::fma(-b, y, 1.0), y, y);
#if KFM_DIV_NR >= 2
    y = std::fma(std::fma(-b, y, 1.0), y, y);
#endif
    const double q = a * y;
    return std::fma(std::fma(-b, q, a), y, q);
}

// 1 / b: seed + one cubically convergent step (<= 1 ulp)
KFM_FN double rcp(double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(b);
#else
    double y = double(1.0f / float(b));
#endif
    const double e = std::fma(-b, y, 1.0);               // |e| <= 2**-24.4: y (1 + e + e^2) is good to e^3 = 2**-73
    return std::fma(y, std::fma(e, e, e), y);            // one cubic step: 3 operations instead of two Newton steps' 4
}

// ---- the libm entry points the scheme uses, on positive finite normal x / moderate arguments ----
KFM_FN double log(double x)
{
    const Split p = split(x);
    return std::fma(p.e, LN2_HI, std::fma(p.e, LN2_LO, ln_mant(p.m)));
}
KFM_FN double log10(double x)
{
    const Split p = split(x);
    return std::fma(p.e, LOG10_2, ln_mant(p.m) * INV_LN10);
}
KFM_FN double exp(double x)                              // |x| < 700
{
    const double n = std::rint(x * INV_LN2);
    const double r = std::fma(-n, LN2_LO, std::fma(-n, LN2_HI, x));
    return std::ldexp(exp_small(r), int(n));
}
KFM_FN double exp10(double x)                            // |x| < 300
{
    const double hi = x * LOG2_10_HI;
    const double lo = std::fma(x, LOG2_10_HI, -hi) + x * LOG2_10_LO;
    const double n = std::rint(hi);
    return std::ldexp(exp2_small((hi - n) + lo), int(n));
}

// ---- general powers: x**y = 2**(y*e + y*log2 m) ----
// libm's pow carries an extended-precision logarithm because the error of log2(x) is multiplied by y and by
// |log2 x| (up to ~40 here).  Splitting off the binary exponent removes that amplification: y*e is formed
// exactly (product + fma residual), |log2 m| <= 1/2, the integer part n of the whole exponent goes to ldexp
// and only the remainder |r| <= 1/2 through the polynomial.  For |y| <= 4.2 (every exponent of the scheme)
// the result is within ~3 ulp of pow.  x must be positive, finite and normal.
struct Log2Parts { double e, lm; };                      // log2(x) = e + lm/ln2, e integral, lm = ln(mantissa)
KFM_FN Log2Parts log2_parts(double x)
{
    const Split p = split(x);
    return Log2Parts{p.e, ln_mant(p.m)};
}
// 2**(t_hi + t_lo + y*log2(x)): t_hi + t_lo is an extra exponent known as an exact sum (0 for a bare power)
KFM_FN double exp2_parts(const Log2Parts &l, double y, double t_hi, double t_lo)
{
    const double p_hi = y * l.e;
    const double p_lo = std::fma(y, l.e, -p_hi);         // y*e = p_hi + p_lo exactly
    const double s = p_hi + t_hi;                        // two-sum: s + s_lo = p_hi + t_hi exactly
    const double bb = s - p_hi;
    const double s_lo = (p_hi - (s - bb)) + (t_hi - bb);
    const double frac = ((s_lo + p_lo) + t_lo) + (y * INV_LN2) * l.lm;   // everything but s, |frac| <= ~2.2
    const double n = std::rint(s + frac);
    const double r = (s - n) + frac;
    return std::ldexp(exp2_small(r), int(n));
}
KFM_FN double pow(double x, double y) { return exp2_parts(log2_parts(x), y, 0., 0.); }
// 10**la * x**y in one exponential (the a_*smo2**b_ pattern of the Field et al. moments, M:1572-1574)
KFM_FN double pow10_times_pow(double la, const Log2Parts &l, double y)
{
    const double t_hi = la * LOG2_10_HI;
    const double t_lo = std::fma(la, LOG2_10_HI, -t_hi) + la * LOG2_10_LO;
    return exp2_parts(l, y, t_hi, t_lo);
}

// ---------------------------------------------------------------------------------------------------------------
// binary32 overloads for the P32n / fp32 instantiations of the column kernel (where the reference declares a
// variable REAL, its Fortran intrinsics are the single-precision ones).  The fp32 special-function units of CDNA
// (v_log_f32, v_exp_f32, v_rsq_f32, v_rcp_f32: ~1 ulp) do the work; arguments are positive, finite and normal as
// above.  Errors: sqrt, cbrt <= 1 ulp; log, log10 <= 2 ulp (absolute error 1e-7 near x = 1, as logf itself is
// evaluated through log2); exp, exp10, pow carry the binary32 rounding of their exponent: relative error about
// 6e-8 * (1 + |exponent in bits|), i.e. <= 3e-6 for the largest exponents of the scheme -- the same order as a
// libm powf on an argument that was itself rounded to binary32.
KFM_FN float log2_hw(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_logf(x);
#else
    return std::log2(x);
#endif
}
KFM_FN float exp2_hw(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x);
#else
    return std::exp2(x);
#endif
}
KFM_FN float log(float x) { return log2_hw(x) * 0.69314718056f; }
KFM_FN float log10(float x) { return log2_hw(x) * 0.30102999566f; }
// e**x = 2**(x*log2 e): the product is formed as hi + lo so that its rounding does not enter the result
KFM_FN float exp(float x)
{
    const float hi = x * 1.44269504089f;
    const float lo = std::fma(x, 1.44269504089f, -hi) + x * 1.925963033e-8f;
    const float n = std::rint(hi);
    return std::ldexp(exp2_hw((hi - n) + lo), int(n));
}
KFM_FN float exp10(float x)
{
    const float hi = x * 3.32192809489f;
    const float lo = std::fma(x, 3.32192809489f, -hi) + x * 6.0576633e-8f;
    const float n = std::rint(hi);
    return std::ldexp(exp2_hw((hi - n) + lo), int(n));
}
KFM_FN float sqrt_pos(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;                               // one Newton step on sqrt: g + (x - g^2) * y/2
    return std::fma(std::fma(-g, g, x), 0.5f * y, g);
#else
    return std::sqrt(x);
#endif
}
KFM_FN float cbrt_pos(float x)
{
    const float y = exp2_hw(-0.33333334f * log2_hw(x));  // x**(-1/3), ~1e-6
    const float t = y * y;
    const float c = x * t;                               // x**(1/3)
    const float r = std::fma(-(c * c), c, x);
    return std::fma(r, t * 0.33333334f, c);              // one Newton correction
}
KFM_FN float pow(float x, float y)
{
    const float l = log2_hw(x);
    const float hi = y * l;
    const float lo = std::fma(y, l, -hi);
    const float n = std::rint(hi);
    return std::ldexp(exp2_hw((hi - n) + lo), int(n));
}
struct Log2PartsF { float l; };
KFM_FN Log2PartsF log2_parts(float x) { return Log2PartsF{log2_hw(x)}; }
KFM_FN float pow10_times_pow(float la, const Log2PartsF &l, float y)
{
    const float a = la * 3.32192809489f, b = y * l.l;
    const float hi = a + b, bb = hi - a;
    const float lo = ((a - (hi - bb)) + (b - bb)) + (std::fma(la, 3.32192809489f, -a) + std::fma(y, l.l, -b));   // two-sum + product residuals
    const float n = std::rint(hi);
    return std::ldexp(exp2_hw((hi - n) + lo), int(n));
}
// mixed-type powers follow Fortran's promotion: REAL**DOUBLE and DOUBLE**REAL are evaluated in DOUBLE
KFM_FN double pow(double x, float y) { return pow(x, double(y)); }
KFM_FN double pow(float x, double y) { return pow(double(x), y); }

}  // namespace fm
}  // namespace kidmp
