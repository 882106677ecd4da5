// thompson_host_init.cpp -- host half of thompson_init (M:374-670): gamma
// constants, rate prefactors, decade offsets, size bins and table axes.  The
// lookup tables themselves are built on the GPU (thompson_tables.hip).
//
// Arithmetic follows the reference's P64 build: NR `gammln` is restated as
// written (M:4598-4620) because every c?g(n) constant inherits its ~2e-10
// accuracy; std::tgamma would NOT reproduce the reference's numbers.
#include "thompson_host_init.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace kidmp {

// ln Gamma(x), 6-term Lanczos exactly as M:4598-4620
static double nr_gammln(double xx)
{
    static const double cof[6] = {76.18009172947146, -86.50532032941677, 24.01409824083091,
                                  -1.231739572450155, .1208650973866179e-2, -.5395239384953e-5};
    const double x = xx;
    double y = x;
    double tmp = x + 5.5;
    tmp = (x + 0.5) * std::log(tmp) - tmp;
    double ser = 1.000000000190015;
    for (double c : cof) {
        y += 1.0;
        ser += c / y;
    }
    return tmp + std::log(2.5066282746310005 * ser / x);
}
static inline double wgamma(double y) { return std::exp(nr_gammln(y)); }   // M:4644-4651

// "1,2,...,9 per decade" axis (M:215-303).  Entries must equal the decimal
// literals of the source, so they are parsed from text rather than multiplied.
static void decade_axis(double *v, int n, int e0)
{
    char buf[24];
    for (int i = 0; i < n; ++i) {
        std::snprintf(buf, sizeof buf, "%de%d", i % 9 + 1, e0 + i / 9);
        v[i] = std::strtod(buf, nullptr);
    }
}

// geometric bins between lo and hi (M:612-669): edges e_n = exp(n/nb*ln(hi/lo)+ln(lo)),
// centre = sqrt(e_n e_{n+1}), width = e_{n+1}-e_n
static void geo_bins(double lo, double hi, double *centre, double *width)
{
    double edge[nbins + 1];
    edge[0] = lo;
    edge[nbins] = hi;
    for (int n = 1; n < nbins; ++n)
        edge[n] = std::exp(double(n) / double(nbins) * std::log(edge[nbins] / edge[0]) + std::log(edge[0]));
    for (int n = 0; n < nbins; ++n) {
        centre[n] = std::sqrt(edge[n] * edge[n + 1]);
        if (width) width[n] = edge[n + 1] - edge[n];
    }
}

void host_init(int iiwarm, int l_sediment, double set_Nc, Consts &c, Bins &b)
{
    c.iiwarm = iiwarm;
    c.l_sediment = l_sediment;
    c.Nt_c = set_Nc * 1.e6;                                   // M:381

    c.Sc3 = std::pow(Sc, 1. / 3.);                            // M:442
    c.D0i = std::pow(xm0i / am_i, 1. / bm_i);                 // M:445
    c.xm0s = am_s * std::pow(D0s, bm_s);
    c.xm0g = am_g * std::pow(D0g, bm_g);

    for (int n = 1; n <= 15; ++n) {                           // M:452-465
        double *e[5] = {&c.cce[0][n - 1], &c.cce[1][n - 1], &c.cce[2][n - 1], &c.cce[3][n - 1], &c.cce[4][n - 1]};
        *e[0] = n + 1.;
        *e[1] = bm_r + n + 1.;
        *e[2] = bm_r + n + 4.;
        *e[3] = n + bv_c + 1.;
        *e[4] = bm_r + n + bv_c + 1.;
        for (int i = 0; i < 5; ++i) c.ccg[i][n - 1] = wgamma(c.cce[i][n - 1]);
        c.ocg1[n - 1] = 1. / c.ccg[0][n - 1];
        c.ocg2[n - 1] = 1. / c.ccg[1][n - 1];
    }

    const double cie_[7] = {mu_i + 1., bm_i + mu_i + 1., bm_i + mu_i + bv_i + 1., mu_i + bv_i + 1.,
                            mu_i + 2., bm_i * 0.5 + mu_i + bv_i + 1., bm_i * 0.5 + mu_i + 1.};   // M:467-473
    for (int n = 0; n < 7; ++n) { c.cie[n] = cie_[n]; c.cig[n] = wgamma(cie_[n]); }
    c.oig1 = 1. / c.cig[0];
    c.oig2 = 1. / c.cig[1];
    c.obmi = 1. / bm_i;

    const double cre_[13] = {bm_r + 1., mu_r + 1., bm_r + mu_r + 1., bm_r * 2. + mu_r + 1.,
                             mu_r + bv_r + 1., bm_r + mu_r + bv_r + 1., bm_r * 0.5 + mu_r + bv_r + 1.,
                             bm_r + mu_r + bv_r + 3., mu_r + bv_r + 3., mu_r + 2.,
                             0.5 * (bv_r + 5. + 2. * mu_r), bm_r * 0.5 + mu_r + 1.,
                             bm_r * 2. + mu_r + bv_r + 1.};                                    // M:485-497
    for (int n = 0; n < 13; ++n) { c.cre[n] = cre_[n]; c.crg[n] = wgamma(cre_[n]); }
    c.obmr = 1. / bm_r;
    c.ore1 = 1. / c.cre[0];
    c.org1 = 1. / c.crg[0];
    c.org2 = 1. / c.crg[1];
    c.org3 = 1. / c.crg[2];

    double cse_[18] = {bm_s + 1., bm_s + 2., bm_s * 2., bm_s + bv_s + 1., bm_s * 2. + bv_s + 1.,
                       bm_s * 2. + 1., bm_s + mu_s + 1., bm_s + mu_s + 2., bm_s + mu_s + 3.,
                       bm_s + mu_s + bv_s + 1., bm_s * 2. + mu_s + bv_s + 1., bm_s * 2. + mu_s + 1.,
                       bv_s + 2., bm_s + bv_s, mu_s + 1., 1.0 + (1.0 + bv_s) / 2., 0., bv_s + mu_s + 3.};
    cse_[16] = cse_[15] + mu_s + 1.;                                                           // M:523
    for (int n = 0; n < 18; ++n) { c.cse[n] = cse_[n]; c.csg[n] = wgamma(cse_[n]); }
    c.oams = 1. / am_s;
    c.obms = 1. / bm_s;
    c.ocms = std::pow(c.oams, c.obms);

    const double cge_[12] = {bm_g + 1., mu_g + 1., bm_g + mu_g + 1., bm_g * 2. + mu_g + 1.,
                             bm_g * 2. + mu_g + bv_g + 1., bm_g + mu_g + bv_g + 1.,
                             bm_g + mu_g + bv_g + 2., bm_g + mu_g + bv_g + 3., mu_g + bv_g + 3.,
                             mu_g + 2., 0.5 * (bv_g + 5. + 2. * mu_g), 0.5 * (bv_g + 5.) + mu_g};   // M:532-543
    for (int n = 0; n < 12; ++n) { c.cge[n] = cge_[n]; c.cgg[n] = wgamma(cge_[n]); }
    c.oamg = 1. / am_g;
    c.obmg = 1. / bm_g;
    c.ocmg = std::pow(c.oamg, c.obmg);
    c.oge1 = 1. / c.cge[0];
    c.ogg1 = 1. / c.cgg[0];
    c.ogg2 = 1. / c.cgg[1];
    c.ogg3 = 1. / c.cgg[2];

    // rate prefactors, M:560-591
    c.t1_qr_qc = PI * .25 * av_r * c.crg[8];
    c.t1_qr_qi = PI * .25 * av_r * c.crg[8];
    c.t2_qr_qi = PI * .25 * am_r * av_r * c.crg[7];
    c.t1_qg_qc = PI * .25 * av_g * c.cgg[8];
    c.t1_qs_qc = PI * .25 * av_s;
    c.t1_qs_qi = PI * .25 * av_s;
    c.t1_qr_ev = 0.78 * c.crg[9];
    c.t2_qr_ev = 0.308 * c.Sc3 * std::sqrt(av_r) * c.crg[10];
    c.t1_qs_sd = 0.86;
    c.t2_qs_sd = 0.28 * c.Sc3 * std::sqrt(av_s);
    c.t1_qs_me = PI * 4. * C_sqrd * olfus * 0.86;
    c.t2_qs_me = PI * 4. * C_sqrd * olfus * 0.28 * c.Sc3 * std::sqrt(av_s);
    c.t1_qg_sd = 0.86 * c.cgg[9];
    c.t2_qg_sd = 0.28 * c.Sc3 * std::sqrt(av_g) * c.cgg[10];
    c.t1_qg_me = PI * 4. * C_cube * olfus * 0.86 * c.cgg[9];
    c.t2_qg_me = PI * 4. * C_cube * olfus * 0.28 * c.Sc3 * std::sqrt(av_g) * c.cgg[10];

    // constant sub-expressions of the solver, evaluated once with the same libm as the reference
    c.lamg_fac = std::pow(c.cgg[2] * c.ogg2 * c.ogg1, c.obmg);
    c.lamr_exp_fac = std::pow(c.crg[2] * c.org2 * c.org1, bm_r);
    c.lamg_exp_fac = std::pow(c.cgg[2] * c.ogg2 * c.ogg1, bm_g);
    for (int n = 0; n < 15; ++n) c.dcg_fac[n] = std::pow(c.ccg[2][n] * c.ocg2[n], c.obmr);
    for (int i = 0; i < 16; ++i) {
        const int n = i < 15 ? i : 14;
        c.lds_tab[0 * 16 + i] = c.ccg[0][n];  c.lds_tab[1 * 16 + i] = c.ccg[1][n];  c.lds_tab[2 * 16 + i] = c.ocg1[n];
        c.lds_tab[3 * 16 + i] = c.ocg2[n];    c.lds_tab[4 * 16 + i] = c.cce[1][n];  c.lds_tab[5 * 16 + i] = c.dcg_fac[n];
        // (ccg(2,n)*ocg1(n))**obmr: rc/(am_r*nc) = ccg(2,n)*ocg1(n)/lamc**3, so the mean-size diameter of M:1693 is this
        // constant over lamc -- the form the reference itself gives Dc_g (M:1699) -- instead of a second cube root
        c.lds_tab[6 * 16 + i] = std::pow(c.ccg[1][n] * c.ocg1[n], c.obmr);
    }
    for (int t = 0; t < 32; ++t) {
        const int n = t - 16, m = n < 0 ? -n : n;
        static const double sq[5] = {10., 100., 1.e4, 1.e8, 1.e16};
        double T = 1.;                                        // 10**m, exact (m <= 16): every partial product is a power of ten
        for (int b = 0; b < 5; ++b) if (m & (1 << b)) T *= sq[b];
        c.lds_tab[112 + t] = n <= 0 ? T : 1. / T;
    }

    // axes, M:215-315
    decade_axis(b.r_c, ntb_c, -6);
    decade_axis(b.r_i, ntb_i, -10);
    decade_axis(b.r_r, ntb_r, -6);
    decade_axis(b.r_g, ntb_g, -5);
    decade_axis(b.r_s, ntb_s, -5);
    decade_axis(b.N0r_exp, ntb_r1, 6);
    decade_axis(b.N0g_exp, ntb_g1, 4);
    decade_axis(b.Nt_i, ntb_i1, 0);
    decade_axis(b.Nt_IN, ntb_IN, 0);
    const double Tc_[ntb_t] = {-0.01, -5., -10., -15., -20., -25., -30., -35., -40.};
    for (int i = 0; i < ntb_t; ++i) b.Tc[i] = Tc_[i];

    // decade offsets, M:594-601
    auto lg = [](double x) { return int(std::lround(std::log10(x))); };
    c.nic2 = lg(b.r_c[0]);
    c.nii2 = lg(b.r_i[0]);
    c.nii3 = lg(b.Nt_i[0]);
    c.nir2 = lg(b.r_r[0]);
    c.nir3 = lg(b.N0r_exp[0]);
    c.nis2 = lg(b.r_s[0]);
    c.nig2 = lg(b.r_g[0]);
    c.nig3 = lg(b.N0g_exp[0]);

    const double sa_[10] = {5.065339, -0.062659, -3.032362, 0.029469, -0.000285,
                            0.31255, 0.000204, 0.003199, 0.0, -0.015952};
    const double sb_[10] = {0.476221, -0.015896, 0.165977, 0.007468, -0.000141,
                            0.060366, 0.000079, 0.000594, 0.0, -0.003577};
    for (int i = 0; i < 10; ++i) { c.sa[i] = sa_[i]; c.sb[i] = sb_[i]; }

    // size bins, M:605-669
    b.Dc[0] = D0c * 1.0;
    b.dtc[0] = D0c * 1.0;
    for (int n = 1; n < nbins; ++n) {
        b.Dc[n] = b.Dc[n - 1] + 1.0e-6;
        b.dtc[n] = b.Dc[n] - b.Dc[n - 1];
    }
    geo_bins(c.D0i * 1.0, 5.0 * D0s, b.Di, b.dti);
    geo_bins(D0r * 1.0, 0.005, b.Dr, b.dtr);
    geo_bins(D0s * 1.0, 0.02, b.Ds, b.dts);
    geo_bins(D0g * 1.0, 0.05, b.Dg, b.dtg);
    geo_bins(1.0, 3000.0, b.t_Nc, nullptr);
    for (int n = 0; n < nbins; ++n) b.t_Nc[n] *= 1.e6;

    c.Dr1 = b.Dr[0];
    c.Drn = b.Dr[nbins - 1];
    c.Ds1 = b.Ds[0];
    c.Dsn = b.Ds[nbins - 1];
    c.r_c1 = b.r_c[0];
    c.r_i1 = b.r_i[0];
    c.r_r1 = b.r_r[0];
    c.r_s1 = b.r_s[0];
    c.r_g1 = b.r_g[0];
    c.Nt_i1 = b.Nt_i[0];
    c.t_Nc1 = b.t_Nc[0];
    c.nic1 = int32_t(std::log(b.t_Nc[nbins - 1] / b.t_Nc[0]));      // INTEGER nic1 truncates DLOG(...) = 7.93, M:195, M:670
    c.pad_ = 0;
}

}  // namespace kidmp
