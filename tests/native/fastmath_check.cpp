// Accuracy check of kid_amd/csrc/fastmath.h on the host (no GPU): max error in ulps against 80-bit long double
// over the argument ranges the Thompson column kernel produces.  Prints one line per function:  name max_ulp
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <random>
#include "fastmath.h"

static double ulp_err(double got, long double want)
{
    const double w = (double)want;
    const double u = std::fabs(std::nextafter(w, INFINITY) - w);
    return (double)(fabsl((long double)got - want) / u);
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0., 1.);
    namespace fm = kidmp::fm;
    double e_sqrt = 0, e_cbrt = 0, e_log = 0, e_log10 = 0, e_exp = 0, e_exp10 = 0, e_pow = 0, e_p10 = 0, e_lm = 0, e_e2 = 0, e_div = 0, e_rcp = 0, e_l1 = 0;
    for (long i = 0; i < n; ++i) {
        // positive normal arguments over 1e-45 .. 1e+25 (mixing ratios, numbers, diameters, slopes)
        const double lx = -45. + 70. * U(rng);
        const double x = std::pow(10., lx) * (1. + U(rng));
        e_log = std::fmax(e_log, ulp_err(fm::log(x), logl((long double)x)));
        e_log10 = std::fmax(e_log10, ulp_err(fm::log10(x), log10l((long double)x)));
        const double xc = std::pow(10., -36. + 72. * U(rng)) * (1. + U(rng));     // cbrt/sqrt domain
        e_cbrt = std::fmax(e_cbrt, ulp_err(fm::cbrt_pos(xc), cbrtl((long double)xc)));
        e_sqrt = std::fmax(e_sqrt, ulp_err(fm::sqrt_pos(xc), sqrtl((long double)xc)));
        const double m = 0.70710678118654757 + U(rng) * (1.4142135623730949 - 0.70710678118654757);
        e_lm = std::fmax(e_lm, ulp_err(fm::ln_mant(m), logl((long double)m)));
        // ln where it vanishes: x within 2**-5 of 1 (and within 1e-9 of it every fourth sample); the RELATIVE error is asked for
        const double x1 = 1. + ((i & 3) ? 0.0625 : 2e-9) * (U(rng) - 0.5);
        e_l1 = std::fmax(e_l1, ulp_err(fm::log(x1), logl((long double)x1)));
        const double r = -0.52 + 1.04 * U(rng);
        e_e2 = std::fmax(e_e2, ulp_err(fm::exp2_small(r), exp2l((long double)r)));
        const double a = -100. + 150. * U(rng);          // exp arguments: -100 .. 50
        e_exp = std::fmax(e_exp, ulp_err(fm::exp(a), expl((long double)a)));
        const double b = -40. + 60. * U(rng);            // exp10 arguments: -40 .. 20
        e_exp10 = std::fmax(e_exp10, ulp_err(fm::exp10(b), powl(10.L, (long double)b)));
        const double y = -4.2 + 8.4 * U(rng);            // the scheme's exponents lie in (-4.2, 4.2)
        e_pow = std::fmax(e_pow, ulp_err(fm::pow(x, y), powl((long double)x, (long double)y)));
        // the kernel's own division and reciprocal (host build: a binary32 seed stands in for v_rcp_f64, 2**-23 vs 2**-24.4)
        const double den = std::pow(10., -30. + 60. * U(rng)) * (1. + U(rng));
        e_div = std::fmax(e_div, ulp_err(fm::div(x, den), (long double)x / (long double)den));
        e_rcp = std::fmax(e_rcp, ulp_err(fm::rcp(den), 1.0L / (long double)den));
        const double la = -12. + 24. * U(rng), yb = 3. * U(rng);
        const double xs = std::pow(10., -12. + 12. * U(rng));
        e_p10 = std::fmax(e_p10, ulp_err(fm::pow10_times_pow(la, fm::log2_parts(xs), yb),
                                        powl(10.L, (long double)la) * powl((long double)xs, (long double)yb)));
    }
    printf("sqrt_pos %.3f\ncbrt_pos %.3f\nln_mant %.3f\nexp2_small %.3f\nlog %.3f\nlog10 %.3f\nexp %.3f\nexp10 %.3f\npow %.3f\npow10_times_pow %.3f\ndiv %.3f\nrcp %.3f\nlog_near_1 %.3f\n",
           e_sqrt, e_cbrt, e_lm, e_e2, e_log, e_log10, e_exp, e_exp10, e_pow, e_p10, e_div, e_rcp, e_l1);
    return 0;
}
