/*
 * thompson_oracle_column.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restates subroutine mp_thompson (M:1156-3688), RSLF/RSIF (M:4656-4717),
 * Eff_aero (M:4354-4390) and the KiD adapter mphys_thompson09_interfacen
 * (W:28-310) in plain C with P64 arithmetic, keeping the reference's block
 * order, loop directions and expression order.  is_aerosol_aware=.false.
 * (M:28), so the aerosol-aware calls (activ_ncloud, iceDeMott, iceKoop,
 * tnc_wev) are unreachable and not restated.
 *
 * Defined-semantics decisions for reference UB:
 *   U1  vtck/vtnck are never assigned (M:3141-3162 commented out) so the
 *       cloud-water sedimentation block M:3414-3425 is undefined; it is
 *       restated as "cloud water does not sediment" (both speeds 0), which
 *       makes the block a no-op on every output.
 *   U4  t_Efrw/t_Efsw second index INT(mvd_c*1e6) is clamped to 1..100
 *       (the reference reads out of bounds beyond).
 */
/*
 * TWO ARITHMETIC MODELS FROM ONE SOURCE.  This file is compiled twice (oracle/Makefile):
 *   P64   (default)   every REAL and DOUBLE PRECISION of the reference is binary64 -- the parity target;
 *   P32n  (-DTH_P32N -fsingle-precision-constant, through thompson_oracle_p32n.c) the reference AS SHIPPED:
 *         every variable the Fortran declares REAL (M:1168-1177 dummies, M:1181-1182 tendencies,
 *         M:1215-1253 work arrays and scalars, RSLF/RSIF, module PARAMETERs M:30-176) is binary32, every
 *         DOUBLE PRECISION one (the ~70 process rates M:1184-1211, ilamr/ilamg/N0_r/N0_g M:1225,
 *         N0_exp/lam_exp/lamc/lamr/lamg/lami/ilami M:1235-1236, the bins Dr/Ds M:206-212, the lookup tables
 *         M:324-340) stays binary64, unsuffixed literals are binary32 like Fortran's, and C's usual
 *         arithmetic conversions then reproduce Fortran's mixed-mode rules (REAL op DOUBLE -> DOUBLE);
 *         <tgmath.h> picks expf/powf/logf... for REAL arguments exactly where the Fortran generic intrinsics
 *         do.  Entry points carry the suffix _p32n and take float arrays.
 */
#include "thompson_oracle_internal.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#ifdef TH_P32N
#include <tgmath.h>
typedef float real;
#define P(name) name##_p32n
#else
typedef double real;
#define P(name) name
#endif

/* real**integer for a REAL base (compiler-rt __powisf2 in P32n, __powidf2 in P64), e.g. 10.**nn at M:1766 */
static inline real th_powi_r(real a, int b)
{
    const int recip = b < 0;
    real r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? (real)1.0 / r : r;
}

/* RSLF M:4656-4686 */
static real rslf(real P, real T)
{
    const real C0 = .611583699E03, C1 = .444606896E02, C2 = .143177157E01,
                 C3 = .264224321E-1, C4 = .299291081E-3, C5 = .203154182E-5,
                 C6 = .702620698E-8, C7 = .379534310E-11, C8 = -.321582393E-13;
    real X = MAXD(-80., T - 273.16);
    real ESL = C0 + X * (C1 + X * (C2 + X * (C3 + X * (C4 + X * (C5 + X * (C6 + X * (C7 + X * C8)))))));
    ESL = MIND(ESL, P * 0.15);
    return .622 * ESL / (P - ESL);
}
/* RSIF M:4691-4717 */
static real rsif(real P, real T)
{
    const real C0 = .609868993E03, C1 = .499320233E02, C2 = .184672631E01,
                 C3 = .402737184E-1, C4 = .565392987E-3, C5 = .521693933E-5,
                 C6 = .307839583E-7, C7 = .105785160E-9, C8 = .161444444E-12;
    real X = MAXD(-80., T - 273.16);
    real ESI = C0 + X * (C1 + X * (C2 + X * (C3 + X * (C4 + X * (C5 + X * (C6 + X * (C7 + X * C8)))))));
    ESI = MIND(ESI, P * 0.15);
    return .622 * ESI / (P - ESI);
}
#ifndef TH_P32N
double th_oracle_rslf(double P_, double T) { return rslf(P_, T); }
double th_oracle_rsif(double P_, double T) { return rsif(P_, T); }
#endif

/* Eff_aero M:4354-4390.  Its results only reach nwfaten/nifaten under
 * is_aerosol_aware (M:2398-2408): dead here, restated for cost fidelity. */
static real Eff_aero(real D, real Da, real visc, real rhoa, real Temp, char species)
{
    const real boltzman = 1.3806503E-23, meanPath = 0.0256E-6;
    real vt = 1.;
    if (species == 'r')
        vt = -0.1021 + 4.932E3 * D - 0.9551E6 * D * D + 0.07934E9 * D * D * D - 0.002362E12 * D * D * D * D;
    else if (species == 's')
        vt = av_s * pow(D, bv_s);
    else if (species == 'g')
        vt = av_g * pow(D, bv_g);
    real Cc = 1. + 2. * meanPath / Da * (1.257 + 0.4 * exp(-0.55 * Da / meanPath));
    real diff = boltzman * Temp * Cc / (3. * PI * visc * Da);
    real Re = 0.5 * rhoa * D * vt / visc;
    real Sc_ = visc / (rhoa * diff);
    real St = Da * Da * vt * 1000. / (9. * visc * D);
    real aval = 1. + log(1. + Re);
    real St2 = (1.2 + 1. / 12. * aval) / (1. + aval);
    real Eff = 4. / (Re * Sc_) * (1. + 0.4 * sqrt(Re) * pow(Sc_, 0.3333) + 0.16 * sqrt(Re) * sqrt(Sc_))
               + 4. * Da / D * (0.02 + Da / D * (1. + 2. * sqrt(Re)));
    if (St > St2) Eff = Eff + pow((St - St2) / (St - St2 + 0.666667), 1.5);
    return MAXD(1.E-5, MIND(Eff, 1.0));
}

/* Field et al. (2005) moment fits, written out at M:1556-1626, M:2673-2711 */
static inline real mom_loga(const real *sa, real tc0, real x)
{
    return sa[1] + sa[2] * tc0 + sa[3] * x + sa[4] * tc0 * x + sa[5] * tc0 * tc0
         + sa[6] * x * x + sa[7] * tc0 * tc0 * x + sa[8] * tc0 * x * x
         + sa[9] * tc0 * tc0 * tc0 + sa[10] * x * x * x;
}
static inline real mom_b(const real *sb, real tc0, real x)
{
    return sb[1] + sb[2] * tc0 + sb[3] * x + sb[4] * tc0 * x + sb[5] * tc0 * tc0
         + sb[6] * x * x + sb[7] * tc0 * tc0 * x + sb[8] * tc0 * x * x
         + sb[9] * tc0 * tc0 * tc0 + sb[10] * x * x * x;
}

/* ---- aerosol-aware branch (is_aerosol_aware = .true.) ---- */
/* iceDeMott M:4720-4759 (DeMott et al. 2010 dust deposition/condensation nucleation) */
static real iceDeMott(real tempc, real qv_, real qvs_, real qvsi_, real rho_, real nifa_)
{
    (void)qv_; (void)qvs_; (void)qvsi_;
    const real rho_not0 = 101325. / (287.05 * 273.15);
    real nifa_cc = nifa_ * rho_not0 * 1.E-6 / rho_;
    real xni = (5.94e-5 * pow(-tempc, (real)3.33)) * pow(nifa_cc, ((-0.0264 * (tempc)) + 0.0033));
    xni = xni * rho_ / rho_not0 * 1000.;
    return MAXD((real)0., xni);
}
/* iceKoop M:4767-4793 (homogeneous freezing of aqueous aerosols, Koop et al. 2001, rate reduced) */
static real iceKoop(real temp_, real qv_, real qvs_, real naero, real DT_)
{
    const real R_uni = 8.314, ar_volume = 4. / 3. * PI * ((real)2.5e-6 * (real)2.5e-6 * (real)2.5e-6);   /* M:155,162: (2.5e-6)**3 */
    real xni = 0.0;
    real satw_ = qv_ / qvs_;
    real mu_diff = 210368.0 + (131.438 * temp_) - (3.32373E6 / temp_) - (41729.1 * log(temp_));
    real a_w_i = exp(mu_diff / (R_uni * temp_));
    real delta_aw = satw_ - a_w_i;
    real log_J_rate = -906.7 + (8502.0 * delta_aw) - (26924.0 * delta_aw * delta_aw) + (29180.0 * delta_aw * delta_aw * delta_aw);
    log_J_rate = MIND((real)20.0, log_J_rate);
    real J_rate = pow((real)10., log_J_rate);
    real prob_h = MIND(1. - exp(-J_rate * ar_volume * DT_), (real)1.);
    if (prob_h > 0.) xni = MIND(prob_h * naero, (real)1000.E3);
    return MAXD((real)0.0, xni);
}
/* activ_ncloud M:4451-4526.  tnccn_act is 1.0 everywhere in this fork (M:752-762: the CCN activation file of the
 * MPAS original is not read), so the bilinear interpolation runs on a table of ones. */
static real activ_ncloud(real Tt, real Ww, real NCCN)
{
    static const real ta_Na[8] = { 0, 10.0, 31.6, 100.0, 316.0, 1000.0, 3160.0, 10000.0 };                 /* ntb_arc = 7 */
    static const real ta_Ww[10] = { 0, 0.01, 0.0316, 0.1, 0.316, 1.0, 3.16, 10.0, 31.6, 100.0 };           /* ntb_arw = 9 */
    const int ntb_arc = 7, ntb_arw = 9;
    real n_local = NCCN * 1.E-6, w_local = Ww;
    int n, i, j;
    (void)Tt;                                                    /* k = table temperature index: every entry is 1.0 */
    if (n_local >= ta_Na[ntb_arc]) n_local = ta_Na[ntb_arc] - 1.0;
    else if (n_local <= ta_Na[1]) n_local = ta_Na[1] + 1.0;
    for (n = 2; n <= ntb_arc; n++) if (n_local >= ta_Na[n - 1] && n_local < ta_Na[n]) break;
    i = n;
    real x1 = log(ta_Na[i - 1]), x2 = log(ta_Na[i]);
    if (w_local >= ta_Ww[ntb_arw]) w_local = ta_Ww[ntb_arw] - 1.0;
    else if (w_local <= ta_Ww[1]) w_local = ta_Ww[1] + 0.001;
    for (n = 2; n <= ntb_arw; n++) if (w_local >= ta_Ww[n - 1] && w_local < ta_Ww[n]) break;
    j = n;
    real y1 = log(ta_Ww[j - 1]), y2 = log(ta_Ww[j]);
    const real A = 1.0, B = 1.0, C = 1.0, D = 1.0;               /* tnccn_act(i-1..i, j-1..j, k, 3, 2) */
    real nx = log(n_local), wy = log(w_local);
    real t = (nx - x1) / (x2 - x1), u = (wy - y1) / (y2 - y1);
    real fraction = (1.0 - t) * (1.0 - u) * A + t * (1.0 - u) * B + t * u * C + (1.0 - t) * u * D;
    return NCCN * fraction;
}

/* "first of {nic-1,nic,nic+1} whose mantissa is in [1,10), else nic+1";
 * idx = INT(x/10**n) + 10*(n-n0) - (n-n0), clamped.  M:1763-1771 and its
 * seven siblings. */
#define DECADE_INDEX(fname, xtype)                                                                     \
static int fname(xtype x, int nic, int n0, int ntb)                                                    \
{                                                                                                      \
    int n = nic - 1;                                                                                   \
    for (int nn = nic - 1; nn <= nic + 1; nn++) {                                                      \
        n = nn;                                                                                        \
        if ((x / th_powi_r(10., nn)) >= 1.0 && (x / th_powi_r(10., nn)) < 10.0) break;                 \
    }                                                                                                  \
    int idx = (int)(x / th_powi_r(10., n)) + 10 * (n - n0) - (n - n0);                                 \
    if (idx > ntb) idx = ntb;                                                                          \
    if (idx < 1) idx = 1;                                                                              \
    return idx;                                                                                        \
}
DECADE_INDEX(decade_index, real)       /* REAL densities: ALOG10 / REAL division, M:1763-1770 ... */
DECADE_INDEX(decade_index_d, double)   /* DOUBLE N0_exp: DLOG10, N0_exp/10.**n with the REAL power promoted, M:1825-1832 */
/* ------------------------------------------------------------------ */
/* What mp_thompson reads from the module (M:25-363, values of thompson_init M:442-602), in the arithmetic of this
 * build: everything the module declares REAL is `real` and is computed here with the statements of M:442-591 in
 * `real` arithmetic (P32n: GAMMLN keeps its DOUBLE PRECISION internals and returns REAL, WGAMMA = EXP(GAMMLN) is a
 * REAL exp, M:4598-4651); the DOUBLE PRECISION bins and the R8 lookup tables are shared with the context.  In the
 * P64 build the values equal the context's own, bit for bit (tests/test_oracle_p32n.py checks that). */
typedef struct {
    int iiwarm, l_sediment, is_aerosol_aware;
    real Nt_c, Sc3, D0i, xm0s, xm0g;
    const double *tnc_wev, *t_Nc; int nic1;
    real sa[11], sb[11];
    real r_c[ntb_c + 1], r_i[ntb_i + 1], r_r[ntb_r + 1], r_g[ntb_g + 1], r_s[ntb_s + 1], Nt_i[ntb_i1 + 1];
    real cce[6][16], ccg[6][16], ocg1[16], ocg2[16];
    real cie[8], cig[8], oig1, oig2, obmi;
    real cre[14], crg[14], ore1, org1, org2, org3, obmr;
    real cse[19], csg[19], oams, obms, ocms;
    real cge[13], cgg[13], oge1, ogg1, ogg2, ogg3, oamg, obmg, ocmg;
    real t1_qr_qc, t1_qr_qi, t2_qr_qi, t1_qg_qc, t1_qs_qc, t1_qs_qi, t1_qr_ev, t2_qr_ev;
    real t1_qs_sd, t2_qs_sd, t1_qg_sd, t2_qg_sd, t1_qs_me, t2_qs_me, t1_qg_me, t2_qg_me;
    int nic2, nii2, nii3, nir2, nir3, nis2, nig2, nig3;
    const double *Dr, *Ds;
    const double *tcg_racg, *tmr_racg, *tcr_gacr, *tmg_gacr, *tnr_racg, *tnr_gacr;
    const double *tcs_racs1, *tmr_racs1, *tcs_racs2, *tmr_racs2, *tcr_sacr1, *tms_sacr1, *tcr_sacr2, *tms_sacr2,
                 *tnr_racs1, *tnr_racs2, *tnr_sacr1, *tnr_sacr2;
    const double *tpi_qcfz, *tni_qcfz, *tpi_qrfz, *tpg_qrfz, *tni_qrfz, *tnr_qrfz;
    const double *tps_iaus, *tni_iaus, *tpi_ide, *t_Efrw, *t_Efsw;
} th_view;

/* GAMMLN M:4598-4620: REAL in, DOUBLE PRECISION inside, REAL out.  WGAMMA M:4644-4651. */
static real gammln_r(real XX)
{
    /* D-exponent literals: the L suffix keeps them out of -fsingle-precision-constant's reach */
    static const double COF[6] = { (double)76.18009172947146e0L, (double)-86.50532032941677e0L, (double)24.01409824083091e0L,
                                   (double)-1.231739572450155e0L, (double).1208650973866179e-2L, (double)-.5395239384953e-5L };
    const double STP = (double)2.5066282746310005e0L;
    double X = XX, Y = X, TMP = X + (double)5.5, SER;
    TMP = (X + (double)0.5) * log(TMP) - TMP;
    SER = (double)1.000000000190015e0L;
    for (int J = 0; J < 6; J++) { Y = Y + (double)1.0; SER = SER + COF[J] / Y; }
    return (real)(TMP + log(STP * SER / X));
}
static real wgamma_r(real y) { return exp(gammln_r(y)); }

void *P(th_oracle_make_view)(const th_oracle *c)
{
    th_view *o = (th_view *)calloc(1, sizeof *o);
    if (!o) return NULL;
    o->iiwarm = c->iiwarm; o->l_sediment = c->l_sediment; o->is_aerosol_aware = c->is_aerosol_aware;
    o->tnc_wev = c->tnc_wev; o->t_Nc = c->t_Nc; o->nic1 = c->nic1;
    o->Nt_c = (real)c->set_Nc * 1.e6;                               /* M:381 */
    for (int i = 1; i <= 10; i++) { o->sa[i] = (real)c->sa[i]; o->sb[i] = (real)c->sb[i]; }   /* REAL PARAMETERs: nearest to the text */
    for (int i = 1; i <= ntb_c; i++) o->r_c[i] = (real)c->r_c[i];
    for (int i = 1; i <= ntb_i; i++) o->r_i[i] = (real)c->r_i[i];
    for (int i = 1; i <= ntb_r; i++) o->r_r[i] = (real)c->r_r[i];
    for (int i = 1; i <= ntb_g; i++) o->r_g[i] = (real)c->r_g[i];
    for (int i = 1; i <= ntb_s; i++) o->r_s[i] = (real)c->r_s[i];
    for (int i = 1; i <= ntb_i1; i++) o->Nt_i[i] = (real)c->Nt_i[i];
    /* M:442-447 */
    o->Sc3 = pow((real)Sc, (real)1. / (real)3.);
    o->D0i = pow((real)xm0i / (real)am_i, (real)1. / (real)bm_i);
    o->xm0s = am_s * pow((real)D0s, (real)bm_s);
    o->xm0g = am_g * pow((real)D0g, (real)bm_g);
    /* M:452-465 */
    for (int n = 1; n <= 15; n++) {
        o->cce[1][n] = n + 1.;
        o->cce[2][n] = bm_r + n + 1.;
        o->cce[3][n] = bm_r + n + 4.;
        o->cce[4][n] = n + bv_c + 1.;
        o->cce[5][n] = bm_r + n + bv_c + 1.;
        for (int i = 1; i <= 5; i++) o->ccg[i][n] = wgamma_r(o->cce[i][n]);
        o->ocg1[n] = 1. / o->ccg[1][n];
        o->ocg2[n] = 1. / o->ccg[2][n];
    }
    /* M:467-483 */
    o->cie[1] = mu_i + 1.;
    o->cie[2] = bm_i + mu_i + 1.;
    o->cie[3] = bm_i + mu_i + bv_i + 1.;
    o->cie[4] = mu_i + bv_i + 1.;
    o->cie[5] = mu_i + 2.;
    o->cie[6] = bm_i * 0.5 + mu_i + bv_i + 1.;
    o->cie[7] = bm_i * 0.5 + mu_i + 1.;
    for (int n = 1; n <= 7; n++) o->cig[n] = wgamma_r(o->cie[n]);
    o->oig1 = 1. / o->cig[1];
    o->oig2 = 1. / o->cig[2];
    o->obmi = 1. / bm_i;
    /* M:485-505 */
    o->cre[1] = bm_r + 1.;
    o->cre[2] = mu_r + 1.;
    o->cre[3] = bm_r + mu_r + 1.;
    o->cre[4] = bm_r * 2. + mu_r + 1.;
    o->cre[5] = mu_r + bv_r + 1.;
    o->cre[6] = bm_r + mu_r + bv_r + 1.;
    o->cre[7] = bm_r * 0.5 + mu_r + bv_r + 1.;
    o->cre[8] = bm_r + mu_r + bv_r + 3.;
    o->cre[9] = mu_r + bv_r + 3.;
    o->cre[10] = mu_r + 2.;
    o->cre[11] = 0.5 * (bv_r + 5. + 2. * mu_r);
    o->cre[12] = bm_r * 0.5 + mu_r + 1.;
    o->cre[13] = bm_r * 2. + mu_r + bv_r + 1.;
    for (int n = 1; n <= 13; n++) o->crg[n] = wgamma_r(o->cre[n]);
    o->obmr = 1. / bm_r;
    o->ore1 = 1. / o->cre[1];
    o->org1 = 1. / o->crg[1];
    o->org2 = 1. / o->crg[2];
    o->org3 = 1. / o->crg[3];
    /* M:507-530 */
    o->cse[1] = bm_s + 1.;
    o->cse[2] = bm_s + 2.;
    o->cse[3] = bm_s * 2.;
    o->cse[4] = bm_s + bv_s + 1.;
    o->cse[5] = bm_s * 2. + bv_s + 1.;
    o->cse[6] = bm_s * 2. + 1.;
    o->cse[7] = bm_s + mu_s + 1.;
    o->cse[8] = bm_s + mu_s + 2.;
    o->cse[9] = bm_s + mu_s + 3.;
    o->cse[10] = bm_s + mu_s + bv_s + 1.;
    o->cse[11] = bm_s * 2. + mu_s + bv_s + 1.;
    o->cse[12] = bm_s * 2. + mu_s + 1.;
    o->cse[13] = bv_s + 2.;
    o->cse[14] = bm_s + bv_s;
    o->cse[15] = mu_s + 1.;
    o->cse[16] = 1.0 + (1.0 + bv_s) / 2.;
    o->cse[17] = o->cse[16] + mu_s + 1.;
    o->cse[18] = bv_s + mu_s + 3.;
    for (int n = 1; n <= 18; n++) o->csg[n] = wgamma_r(o->cse[n]);
    o->oams = 1. / am_s;
    o->obms = 1. / bm_s;
    o->ocms = pow(o->oams, o->obms);
    /* M:532-553 */
    o->cge[1] = bm_g + 1.;
    o->cge[2] = mu_g + 1.;
    o->cge[3] = bm_g + mu_g + 1.;
    o->cge[4] = bm_g * 2. + mu_g + 1.;
    o->cge[5] = bm_g * 2. + mu_g + bv_g + 1.;
    o->cge[6] = bm_g + mu_g + bv_g + 1.;
    o->cge[7] = bm_g + mu_g + bv_g + 2.;
    o->cge[8] = bm_g + mu_g + bv_g + 3.;
    o->cge[9] = mu_g + bv_g + 3.;
    o->cge[10] = mu_g + 2.;
    o->cge[11] = 0.5 * (bv_g + 5. + 2. * mu_g);
    o->cge[12] = 0.5 * (bv_g + 5.) + mu_g;
    for (int n = 1; n <= 12; n++) o->cgg[n] = wgamma_r(o->cge[n]);
    o->oamg = 1. / am_g;
    o->obmg = 1. / bm_g;
    o->ocmg = pow(o->oamg, o->obmg);
    o->oge1 = 1. / o->cge[1];
    o->ogg1 = 1. / o->cgg[1];
    o->ogg2 = 1. / o->cgg[2];
    o->ogg3 = 1. / o->cgg[3];
    /* rate prefactors M:560-591 */
    o->t1_qr_qc = PI * .25 * av_r * o->crg[9];
    o->t1_qr_qi = PI * .25 * av_r * o->crg[9];
    o->t2_qr_qi = PI * .25 * am_r * av_r * o->crg[8];
    o->t1_qg_qc = PI * .25 * av_g * o->cgg[9];
    o->t1_qs_qc = PI * .25 * av_s;
    o->t1_qs_qi = PI * .25 * av_s;
    o->t1_qr_ev = 0.78 * o->crg[10];
    o->t2_qr_ev = 0.308 * o->Sc3 * sqrt((real)av_r) * o->crg[11];
    o->t1_qs_sd = 0.86;
    o->t2_qs_sd = 0.28 * o->Sc3 * sqrt((real)av_s);
    o->t1_qs_me = PI * 4. * C_sqrd * olfus * 0.86;
    o->t2_qs_me = PI * 4. * C_sqrd * olfus * 0.28 * o->Sc3 * sqrt((real)av_s);
    o->t1_qg_sd = 0.86 * o->cgg[10];
    o->t2_qg_sd = 0.28 * o->Sc3 * sqrt((real)av_g) * o->cgg[11];
    o->t1_qg_me = PI * 4. * C_cube * olfus * 0.86 * o->cgg[10];
    o->t2_qg_me = PI * 4. * C_cube * olfus * 0.28 * o->Sc3 * sqrt((real)av_g) * o->cgg[11];
    /* M:594-602 (the same integers in either arithmetic) */
    o->nic2 = c->nic2; o->nii2 = c->nii2; o->nii3 = c->nii3; o->nir2 = c->nir2; o->nir3 = c->nir3;
    o->nis2 = c->nis2; o->nig2 = c->nig2; o->nig3 = c->nig3;
    o->Dr = c->Dr; o->Ds = c->Ds;
    o->tcg_racg = c->tcg_racg; o->tmr_racg = c->tmr_racg; o->tcr_gacr = c->tcr_gacr; o->tmg_gacr = c->tmg_gacr;
    o->tnr_racg = c->tnr_racg; o->tnr_gacr = c->tnr_gacr;
    o->tcs_racs1 = c->tcs_racs1; o->tmr_racs1 = c->tmr_racs1; o->tcs_racs2 = c->tcs_racs2; o->tmr_racs2 = c->tmr_racs2;
    o->tcr_sacr1 = c->tcr_sacr1; o->tms_sacr1 = c->tms_sacr1; o->tcr_sacr2 = c->tcr_sacr2; o->tms_sacr2 = c->tms_sacr2;
    o->tnr_racs1 = c->tnr_racs1; o->tnr_racs2 = c->tnr_racs2; o->tnr_sacr1 = c->tnr_sacr1; o->tnr_sacr2 = c->tnr_sacr2;
    o->tpi_qcfz = c->tpi_qcfz; o->tni_qcfz = c->tni_qcfz; o->tpi_qrfz = c->tpi_qrfz; o->tpg_qrfz = c->tpg_qrfz;
    o->tni_qrfz = c->tni_qrfz; o->tnr_qrfz = c->tnr_qrfz; o->tps_iaus = c->tps_iaus; o->tni_iaus = c->tni_iaus;
    o->tpi_ide = c->tpi_ide; o->t_Efrw = c->t_Efrw; o->t_Efsw = c->t_Efsw;
    return o;
}

/* the view's constants by name, for tests (P64: must equal the context's; P32n: the fp32 values) */
double P(th_oracle_view_const)(const th_oracle *c, const char *name, int idx)
{
    const th_view *o = (const th_view *)c->P(view);
#define VC(n) if (!strcmp(name, #n)) return (double)o->n
#define VA(n) if (!strcmp(name, #n)) return (double)o->n[idx]
    VC(Nt_c); VC(Sc3); VC(D0i); VC(xm0s); VC(xm0g); VC(oig1); VC(oig2); VC(org1); VC(org2); VC(org3); VC(oams);
    VC(ocms); VC(ocmg); VC(ogg1); VC(ogg2); VC(ogg3); VC(t1_qr_qc); VC(t2_qr_qi); VC(t1_qg_qc); VC(t1_qr_ev); VC(t2_qr_ev);
    VC(t2_qs_sd); VC(t1_qs_me); VC(t2_qs_me); VC(t1_qg_sd); VC(t2_qg_sd); VC(t1_qg_me); VC(t2_qg_me);
    VA(cie); VA(cig); VA(cre); VA(crg); VA(cse); VA(csg); VA(cge); VA(cgg); VA(ocg1); VA(ocg2);
    if (!strcmp(name, "ccg1")) return (double)o->ccg[1][idx];
    if (!strcmp(name, "ccg2")) return (double)o->ccg[2][idx];
    if (!strcmp(name, "ccg3")) return (double)o->ccg[3][idx];
#undef VC
#undef VA
    return -1.0e30;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

#ifndef TH_P32N
int th_oracle_mp_thompson(const th_oracle *o,
                          double *qv1d, double *qc1d, double *qi1d,
                          double *qr1d, double *qs1d, double *qg1d,
                          double *ni1d, double *nr1d, double *nc1d,
                          double *nwfa1d, double *nifa1d, double *t1d,
                          const double *p1d, const double *w1d,
                          const double *dzq, double ppt[4],
                          int nz, double dt, double *rates, int *nstep_out)
{
    return th_oracle_mp_thompson_force(o, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d, nifa1d,
                                       t1d, p1d, w1d, dzq, ppt, nz, dt, rates, nstep_out, NULL, 0);
}

int th_oracle_mp_thompson_ex(const th_oracle *o,
                             double *qv1d, double *qc1d, double *qi1d,
                             double *qr1d, double *qs1d, double *qg1d,
                             double *ni1d, double *nr1d, double *nc1d,
                             double *nwfa1d, double *nifa1d, double *t1d,
                             const double *p1d, const double *w1d,
                             const double *dzq, double ppt[4],
                             int nz, double dt, double *rates, int *nstep_out, int *illcond)
{
    return th_oracle_mp_thompson_force(o, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d, nifa1d,
                                       t1d, p1d, w1d, dzq, ppt, nz, dt, rates, nstep_out, illcond, 0);
}

#endif

/* mp_thompson, M:1156-3688.  Dummies are REAL (M:1168-1177); rates (oracle-only output) are the DOUBLE PRECISION
 * process rates as passed to save_dg (M:2967-3119). */
int P(th_oracle_mp_thompson_force)(const th_oracle *ctx,
                                real *qv1d, real *qc1d, real *qi1d,
                                real *qr1d, real *qs1d, real *qg1d,
                                real *ni1d, real *nr1d, real *nc1d,
                                real *nwfa1d, real *nifa1d, real *t1d,
                                const real *p1d, const real *w1d,
                                const real *dzq, real ppt[4],
                                int nz, real dt, double *rates, int *nstep_out, int *illcond, int force)
{
    const th_view *o = (const th_view *)ctx->P(view);
    const int aero = o->is_aerosol_aware;                      /* M:28; .false. in KiD, settable here */
    const int kts = 0, kte = nz - 1;
    const size_t NA = 100, ND = 80;
    real *ws = (real *)calloc(NA * (size_t)(nz + 2), sizeof(real));          /* REAL work arrays */
    double *wd = (double *)calloc(ND * (size_t)(nz + 2), sizeof(double));    /* DOUBLE PRECISION work arrays */
    int *Lws = (int *)calloc(5 * (size_t)nz, sizeof(int));
    if (!ws || !wd || !Lws) { free(ws); free(wd); free(Lws); return -1; }
    size_t wi = 0, di = 0;
#define A(name) real *name = ws + (wi++) * (size_t)(nz + 2)
#define AD(name) double *name = wd + (di++) * (size_t)(nz + 2)
    /* tendencies M:1181-1182 */
    A(tten); A(qvten); A(qcten); A(qiten); A(qrten); A(qsten); A(qgten);
    A(niten); A(nrten); A(ncten); A(nwfaten); A(nifaten);
    /* rates M:1184-1211 */
    AD(prw_vcd); AD(pnc_wcd); AD(pnc_wau); AD(pnc_rcw); AD(pnc_scw); AD(pnc_gcw);
    AD(pna_rca); AD(pna_sca); AD(pna_gca); AD(pnd_rcd); AD(pnd_scd); AD(pnd_gcd);
    AD(prr_wau); AD(prr_rcw); AD(prr_rcs); AD(prr_rcg); AD(prr_sml); AD(prr_gml);
    AD(prr_rci); AD(prv_rev); AD(pnr_wau); AD(pnr_rcs); AD(pnr_rcg); AD(pnr_rci);
    AD(pnr_sml); AD(pnr_gml); AD(pnr_rev); AD(pnr_rcr); AD(pnr_rfz);
    AD(pri_inu); AD(pni_inu); AD(pri_ihm); AD(pni_ihm); AD(pri_wfz); AD(pni_wfz);
    AD(pri_rfz); AD(pni_rfz); AD(pri_ide); AD(pni_ide); AD(pri_rci); AD(pni_rci);
    AD(pni_sci); AD(pni_iau); AD(pri_iha); AD(pni_iha);
    AD(prs_iau); AD(prs_sci); AD(prs_rcs); AD(prs_scw); AD(prs_sde); AD(prs_ihm); AD(prs_ide);
    AD(prg_scw); AD(prg_rfz); AD(prg_gde); AD(prg_gcw); AD(prg_rci); AD(prg_rcs); AD(prg_rcg); AD(prg_ihm);
    /* state M:1215-1241 */
    A(temp); A(pres); A(qv);
    A(rc); A(ri); A(rr); A(rs); A(rg); A(ni); A(nr); A(nc); A(nwfa); A(nifa);
    A(rho); A(rhof); A(rhof2);
    A(qvs); A(qvsi); A(delQvs);
    A(satw); A(sati); A(ssatw); A(ssati);
    A(diffu); A(visco); A(vsc2); A(tcond); A(lvap); A(ocp); A(lvt2);
    AD(ilamr); AD(ilamg); AD(N0_r); AD(N0_g);          /* DOUBLE PRECISION, M:1225 */
    A(mvd_r); A(mvd_c);
    A(smob); A(smo2); A(smo1); A(smo0); A(smoc); A(smod); A(smoe); A(smof);
    A(sed_r); A(sed_s); A(sed_g); A(sed_i); A(sed_n);
    A(vtik); A(vtnik); A(vtrk); A(vtnrk); A(vtsk); A(vtgk);
    A(vts_boost);
#undef A
#undef AD
    int *L_qc = Lws, *L_qi = Lws + nz, *L_qr = Lws + 2 * nz, *L_qs = Lws + 3 * nz, *L_qg = Lws + 4 * nz;

    const real *cce2 = o->cce[2], *ccg1 = o->ccg[1], *ccg2 = o->ccg[2], *ccg3 = o->ccg[3];
    const real *ocg1 = o->ocg1, *ocg2 = o->ocg2;
    const real *cie = o->cie, *cig = o->cig, *cre = o->cre, *crg = o->crg,
                 *cse = o->cse, *csg = o->csg, *cge = o->cge, *cgg = o->cgg;
    const real oig1 = o->oig1, oig2 = o->oig2, obmi = o->obmi, obmr = o->obmr,
                 org1 = o->org1, org2 = o->org2, org3 = o->org3, oams = o->oams,
                 oge1 = o->oge1, ogg1 = o->ogg1, ogg2 = o->ogg2, ogg3 = o->ogg3, obmg = o->obmg;
    const real Nt_c = o->Nt_c, D0i = o->D0i;
    const int iiwarm = o->iiwarm;
    const real DT = dt;

    /* scalar locals with the types of M:1233-1253 */
    real rgvm, delta_tp, orho, lfus2;
    real onstep[6];
    double N0_exp, N0_min, lam_exp, lamc = 0., lamr, lamg;      /* DOUBLE PRECISION, M:1235 */
    double lami, ilami;                                         /* DOUBLE PRECISION, M:1236 */
    real xDc = 0., Dc_b, Dc_g, xDi, xDs, xDg;
    real zeta1, zeta, taud, tau;
    real stoke_g;
    real vti, vtr, vts, vtg;
    real Mrat, ils1, ils2, t1_vts, t2_vts, t3_vts, t4_vts, C_snow;
    real a_, b_, loga_, tf;
    real tempc = 0., tc0;
    real xnc, xri, xni, xmi, oxmi, xrc, xrr, xnr;
    real xsat, rate_max, sump, ratio;                           /* M:1246: REAL even though they scale DOUBLE rates */
    real clap, fcd, dfcd;
    real otemp, rvs, rvs_p, rvs_pp, gamsc, alphsc, t1_evap, t1_subl;
    real r_frac, g_frac;
    real Ef_rw, Ef_sw, Ef_gw = 0., Ef_rr, Ef_ra, Ef_sa, Ef_ga;
    real dtsave, odts, odt, odzq;
    real xslw1, ygra1, zans1, eva_factor;
    int k, n, nstep, k_0, idx, nu_c = 12;
    int ksed1[6];
    int idx_tc, idx_t, idx_s, idx_g1, idx_g, idx_r1, idx_r, idx_i1, idx_i, idx_c;
    int no_micro;
    (void)rgvm; (void)Ef_ra; (void)Ef_sa; (void)Ef_ga;

    no_micro = 1;                                           /* M:1276-1280 */
    dtsave = dt;
    odt = 1. / dt;
    odts = 1. / dtsave;
    /* M:1282-1381: all arrays are zero from calloc */
    if (rates) memset(rates, 0, sizeof(double) * TH_ORACLE_NRATES * (size_t)nz);
    if (nstep_out) { nstep_out[0] = nstep_out[1] = nstep_out[2] = nstep_out[3] = 0; }
    if (illcond) memset(illcond, 0, sizeof(int) * (size_t)nz);

    /* ---- B: put column of data into local arrays, M:1387-1493 ---- */
    for (k = kts; k <= kte; k++) {
        temp[k] = t1d[k];
        qv[k] = MAXD(1.E-10, qv1d[k]);
        pres[k] = p1d[k];
        rho[k] = 0.622 * pres[k] / (R_gas * temp[k] * (qv[k] + 0.622));
        nwfa[k] = MAXD(11.1E6, MIND(9999.E6, nwfa1d[k] * rho[k]));
        nifa[k] = MAXD(naIN1 * 0.01, MIND(9999.E6, nifa1d[k] * rho[k]));

        if (qc1d[k] > R1) {
            no_micro = 0;
            rc[k] = qc1d[k] * rho[k];
            nc[k] = MAXD(2., nc1d[k] * rho[k]);
            L_qc[k] = 1;
            nu_c = NINT(1000.E6 / nc[k]) + 2; if (nu_c > 15) nu_c = 15;
            lamc = pow(nc[k] * am_r * ccg2[nu_c] * ocg1[nu_c] / rc[k], obmr);
            xDc = (bm_r + nu_c + 1.) / lamc;
            if (xDc < D0c)
                lamc = cce2[nu_c] / D0c;
            else if (xDc > D0r * 2.)
                lamc = cce2[nu_c] / (D0r * 2.);
            nc[k] = MIND(Nt_c_max, ccg1[nu_c] * ocg2[nu_c] * rc[k] / am_r * pow(lamc, bm_r));
            if (!aero) nc[k] = Nt_c;                         /* M:1410 */
        } else {
            qc1d[k] = 0.0;
            nc1d[k] = 0.0;
            rc[k] = R1;
            nc[k] = 2.;
            L_qc[k] = 0;
        }

        if (qi1d[k] > R1) {
            no_micro = 0;
            ri[k] = qi1d[k] * rho[k];
            ni[k] = MAXD(R2, ni1d[k] * rho[k]);
            if (ni[k] <= R2) {
                lami = cie[2] / 25.E-6;
                ni[k] = MIND(499.e3, cig[1] * oig2 * ri[k] / am_i * pow(lami, bm_i));
            }
            L_qi[k] = 1;
            lami = pow(am_i * cig[2] * oig1 * ni[k] / ri[k], obmi);
            ilami = 1. / lami;
            xDi = (bm_i + mu_i + 1.) * ilami;
            if (xDi < 5.E-6) {
                lami = cie[2] / 5.E-6;
                ni[k] = MIND(499.e3, cig[1] * oig2 * ri[k] / am_i * pow(lami, bm_i));
            } else if (xDi > 300.E-6) {
                lami = cie[2] / 300.E-6;
                ni[k] = cig[1] * oig2 * ri[k] / am_i * pow(lami, bm_i);
            }
        } else {
            qi1d[k] = 0.0;
            ni1d[k] = 0.0;
            ri[k] = R1;
            ni[k] = R2;
            L_qi[k] = 0;
        }

        if (qr1d[k] > R1) {
            no_micro = 0;
            rr[k] = qr1d[k] * rho[k];
            nr[k] = MAXD(R2, nr1d[k] * rho[k]);
            if (nr[k] <= R2) {
                mvd_r[k] = 1.0E-3;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                nr[k] = crg[2] * org3 * rr[k] * pow(lamr, bm_r) / am_r;
            }
            L_qr[k] = 1;
            lamr = pow(am_r * crg[3] * org2 * nr[k] / rr[k], obmr);
            mvd_r[k] = (3.0 + mu_r + 0.672) / lamr;
            if (mvd_r[k] > 2.5E-3) {
                mvd_r[k] = 2.5E-3;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                nr[k] = crg[2] * org3 * rr[k] * pow(lamr, bm_r) / am_r;
            } else if (mvd_r[k] < D0r * 0.75) {
                mvd_r[k] = D0r * 0.75;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                nr[k] = crg[2] * org3 * rr[k] * pow(lamr, bm_r) / am_r;
            }
        } else {
            qr1d[k] = 0.0;
            nr1d[k] = 0.0;
            rr[k] = R1;
            nr[k] = R2;
            L_qr[k] = 0;
        }
        if (qs1d[k] > R1) {
            no_micro = 0;
            rs[k] = qs1d[k] * rho[k];
            L_qs[k] = 1;
        } else {
            qs1d[k] = 0.0;
            rs[k] = R1;
            L_qs[k] = 0;
        }
        if (qg1d[k] > R1) {
            no_micro = 0;
            rg[k] = qg1d[k] * rho[k];
            L_qg[k] = 1;
        } else {
            qg1d[k] = 0.0;
            rg[k] = R1;
            L_qg[k] = 0;
        }
    }

    /* ---- C: thermodynamics, M:1503-1533 ---- */
    for (k = kts; k <= kte; k++) {
        tempc = temp[k] - 273.15;
        rhof[k] = sqrt(rho_not / rho[k]);
        rhof2[k] = sqrt(rhof[k]);
        qvs[k] = rslf(pres[k], temp[k]);
        delQvs[k] = MAXD(0.0, rslf(pres[k], 273.15) - qv[k]);
        if (tempc <= 0.0)
            qvsi[k] = rsif(pres[k], temp[k]);
        else
            qvsi[k] = qvs[k];
        satw[k] = qv[k] / qvs[k];
        sati[k] = qv[k] / qvsi[k];
        ssatw[k] = satw[k] - 1.;
        ssati[k] = sati[k] - 1.;
        if (fabs(ssatw[k]) < eps) ssatw[k] = 0.0;
        if (fabs(ssati[k]) < eps) ssati[k] = 0.0;
        if (no_micro && ssati[k] > 0.0) no_micro = 0;
        diffu[k] = 2.11E-5 * pow(temp[k] / 273.15, 1.94) * (101325. / pres[k]);
        if (tempc >= 0.0)
            visco[k] = (1.718 + 0.0049 * tempc) * 1.0E-5;
        else
            visco[k] = (1.718 + 0.0049 * tempc - 1.2E-5 * tempc * tempc) * 1.0E-5;
        ocp[k] = 1. / (Cp * (1. + 0.887 * qv[k]));
        vsc2[k] = sqrt(rho[k] / visco[k]);
        lvap[k] = lvap0 + (2106.0 - 4218.0) * tempc;
        tcond[k] = (5.69 + 0.0168 * tempc) * 1.0E-5 * 418.936;
    }

    if (no_micro) { free(ws); free(wd); free(Lws); return 1; }          /* M:1540 */

    /* ---- D: snow moments, M:1545-1628 ---- */
    if (!iiwarm) {
        for (k = kts; k <= kte; k++) {
            if (!L_qs[k]) continue;
            tc0 = MIND(-0.1, temp[k] - 273.15);
            smob[k] = rs[k] * oams;
            if (bm_s > (2.0 - 1.e-3) && bm_s < (2.0 + 1.e-3)) {
                smo2[k] = smob[k];
            } else {
                loga_ = mom_loga(o->sa, tc0, bm_s);
                a_ = pow(10.0, loga_);
                b_ = mom_b(o->sb, tc0, bm_s);
                smo2[k] = pow(smob[k] / a_, 1. / b_);
            }
            /* 0th moment M:1571-1574 */
            loga_ = o->sa[1] + o->sa[2] * tc0 + o->sa[5] * tc0 * tc0 + o->sa[9] * tc0 * tc0 * tc0;
            a_ = pow(10.0, loga_);
            b_ = o->sb[1] + o->sb[2] * tc0 + o->sb[5] * tc0 * tc0 + o->sb[9] * tc0 * tc0 * tc0;
            smo0[k] = a_ * pow(smo2[k], b_);
            /* 1st moment M:1577-1587 */
            loga_ = o->sa[1] + o->sa[2] * tc0 + o->sa[3] + o->sa[4] * tc0 + o->sa[5] * tc0 * tc0
                  + o->sa[6] + o->sa[7] * tc0 * tc0 + o->sa[8] * tc0 + o->sa[9] * tc0 * tc0 * tc0 + o->sa[10];
            a_ = pow(10.0, loga_);
            b_ = o->sb[1] + o->sb[2] * tc0 + o->sb[3] + o->sb[4] * tc0 + o->sb[5] * tc0 * tc0
               + o->sb[6] + o->sb[7] * tc0 * tc0 + o->sb[8] * tc0 + o->sb[9] * tc0 * tc0 * tc0 + o->sb[10];
            smo1[k] = a_ * pow(smo2[k], b_);
            /* bm_s+1 M:1590-1600 */
            loga_ = mom_loga(o->sa, tc0, cse[1]);
            a_ = pow(10.0, loga_);
            b_ = mom_b(o->sb, tc0, cse[1]);
            smoc[k] = a_ * pow(smo2[k], b_);
            /* bv_s+2 M:1603-1613 */
            loga_ = mom_loga(o->sa, tc0, cse[13]);
            a_ = pow(10.0, loga_);
            b_ = mom_b(o->sb, tc0, cse[13]);
            smoe[k] = a_ * pow(smo2[k], b_);
            /* 1+(bv_s+1)/2 M:1616-1626 */
            loga_ = mom_loga(o->sa, tc0, cse[16]);
            a_ = pow(10.0, loga_);
            b_ = mom_b(o->sb, tc0, cse[16]);
            smof[k] = a_ * pow(smo2[k], b_);
        }

        /* ---- E: graupel intercept/slope, M:1633-1654 ---- */
        N0_min = gonv_max;
        k_0 = kts;
        for (k = kte; k >= kts; k--)
            if (temp[k] >= 270.65) k_0 = k_0 > k ? k_0 : k;
        for (k = kte; k >= kts; k--) {
            if (k > k_0 && L_qr[k] && mvd_r[k] > 100.E-6)
                xslw1 = 4.01 + log10(mvd_r[k]);
            else
                xslw1 = 0.01;
            ygra1 = 4.31 + log10(MAXD(5.E-5, rg[k]));
            zans1 = 3.1 + (100. / (300. * xslw1 * ygra1 / (10. / xslw1 + 1. + 0.25 * ygra1) + 30. + 10. * ygra1));
            N0_exp = pow(10., zans1);
            N0_exp = MAXD(gonv_min, MIND(N0_exp, gonv_max));
            N0_min = MIND(N0_exp, N0_min);
            N0_exp = N0_min;
            lam_exp = pow(N0_exp * am_g * cgg[1] / rg[k], oge1);
            lamg = lam_exp * pow(cgg[3] * ogg2 * ogg1, obmg);
            ilamg[k] = 1. / lamg;
            N0_g[k] = N0_exp / (cgg[2] * lam_exp) * pow(lamg, cge[2]);
        }
    }

    /* ---- F: rain slope/intercept, M:1661-1666 ---- */
    for (k = kte; k >= kts; k--) {
        lamr = pow(am_r * crg[3] * org2 * nr[k] / rr[k], obmr);
        ilamr[k] = 1. / lamr;
        mvd_r[k] = (3.0 + mu_r + 0.672) / lamr;
        N0_r[k] = nr[k] * org2 * pow(lamr, cre[2]);
    }

    /* ---- G: warm-rain process terms, M:1676-1742 ---- */
    for (k = kts; k <= kte; k++) {
        if (L_qr[k] && mvd_r[k] > D0r) {
            Ef_rr = 1.0 - exp(2300.0 * (mvd_r[k] - 1950.0E-6));
            pnr_rcr[k] = Ef_rr * 2.0 * nr[k] * rr[k];
        }

        mvd_c[k] = D0c;
        if (L_qc[k]) {
            nu_c = NINT(1000.E6 / nc[k]) + 2; if (nu_c > 15) nu_c = 15;
            xDc = MAXD(D0c * 1.E6, pow(rc[k] / (am_r * nc[k]), obmr) * 1.E6);
            lamc = pow(nc[k] * am_r * ccg2[nu_c] * ocg1[nu_c] / rc[k], obmr);
            mvd_c[k] = (3.0 + nu_c + 0.672) / lamc;
        }

        if (rc[k] > 0.01e-3) {
            Dc_g = (pow(ccg3[nu_c] * ocg2[nu_c], obmr) / lamc) * 1.E6;
            Dc_b = pow(xDc * xDc * xDc * Dc_g * Dc_g * Dc_g - xDc * xDc * xDc * xDc * xDc * xDc, 1. / 6.);
            zeta1 = 0.5 * ((6.25E-6 * xDc * Dc_b * Dc_b * Dc_b - 0.4) + fabs(6.25E-6 * xDc * Dc_b * Dc_b * Dc_b - 0.4));
            zeta = 0.027 * rc[k] * zeta1;
            taud = 0.5 * ((0.5 * Dc_b - 7.5) + fabs(0.5 * Dc_b - 7.5)) + R1;
            tau = 3.72 / (rc[k] * taud);
            prr_wau[k] = zeta / tau;
            prr_wau[k] = MIND(rc[k] * odts, prr_wau[k]);
            pnr_wau[k] = prr_wau[k] / (am_r * nu_c * D0r * D0r * D0r);
            pnc_wau[k] = MIND(nc[k] * odts, prr_wau[k] / (am_r * mvd_c[k] * mvd_c[k] * mvd_c[k]));
        }

        if (L_qr[k] && mvd_r[k] > D0r && mvd_c[k] > D0c) {
            lamr = 1. / ilamr[k];
            idx = 1 + (int)(nbr * log(mvd_r[k] / o->Dr[1]) / log(o->Dr[nbr] / o->Dr[1]));
            idx = idx < nbr ? idx : nbr;
            if (idx < 1) idx = 1;
            Ef_rw = EFRW(o->t_Efrw, idx, clampi((int)(mvd_c[k] * 1.E6), 1, nbc));
            prr_rcw[k] = rhof[k] * o->t1_qr_qc * Ef_rw * rc[k] * N0_r[k] * pow(lamr + fv_r, -cre[9]);
            prr_rcw[k] = MIND(rc[k] * odts, prr_rcw[k]);
            pnc_rcw[k] = rhof[k] * o->t1_qr_qc * Ef_rw * nc[k] * N0_r[k] * pow(lamr + fv_r, -cre[9]);
            pnc_rcw[k] = MIND(nc[k] * odts, pnc_rcw[k]);
        }

        if (L_qr[k] && mvd_r[k] > D0r) {                    /* dead outputs, M:1729-1740 */
            Ef_ra = Eff_aero(mvd_r[k], 0.04E-6, visco[k], rho[k], temp[k], 'r');
            lamr = 1. / ilamr[k];
            pna_rca[k] = rhof[k] * o->t1_qr_qc * Ef_ra * nwfa[k] * N0_r[k] * pow(lamr + fv_r, -cre[9]);
            pna_rca[k] = MIND(nwfa[k] * odts, pna_rca[k]);
            Ef_ra = Eff_aero(mvd_r[k], 0.8E-6, visco[k], rho[k], temp[k], 'r');
            pnd_rcd[k] = rhof[k] * o->t1_qr_qc * Ef_ra * nifa[k] * N0_r[k] * pow(lamr + fv_r, -cre[9]);
            pnd_rcd[k] = MIND(nifa[k] * odts, pnd_rcd[k]);
        }
    }

    /* ---- H: frozen-species process terms, M:1749-2286 ---- */
    if (!iiwarm) {
        for (k = kts; k <= kte; k++) {
            vts_boost[k] = 1.5;

            tempc = temp[k] - 273.15;
            idx_tc = NINT(-tempc); if (idx_tc > 45) idx_tc = 45; if (idx_tc < 1) idx_tc = 1;
            idx_t = (int)((tempc - 2.5) / 5.) - 1;
            idx_t = 1 > -idx_t ? 1 : -idx_t;
            idx_t = idx_t < ntb_t ? idx_t : ntb_t;
            /* IT (M:1759) and idx_n (M:1777-1778) feed nothing */

            if (rc[k] > o->r_c[1])
                idx_c = decade_index(rc[k], NINT(log10(rc[k])), o->nic2, ntb_c);
            else
                idx_c = 1;

            if (ri[k] > o->r_i[1])
                idx_i = decade_index(ri[k], NINT(log10(ri[k])), o->nii2, ntb_i);
            else
                idx_i = 1;

            if (ni[k] > o->Nt_i[1])
                idx_i1 = decade_index(ni[k], NINT(log10(ni[k])), o->nii3, ntb_i1);
            else
                idx_i1 = 1;

            if (rr[k] > o->r_r[1]) {
                idx_r = decade_index(rr[k], NINT(log10(rr[k])), o->nir2, ntb_r);
                lamr = 1. / ilamr[k];
                lam_exp = lamr * pow(crg[3] * org2 * org1, bm_r);
                N0_exp = org1 * rr[k] / am_r * pow(lam_exp, cre[1]);
                idx_r1 = decade_index_d(N0_exp, NINT(log10(N0_exp)), o->nir3, ntb_r1);
            } else {
                idx_r = 1;
                idx_r1 = ntb_r1;
            }

            if (rs[k] > o->r_s[1])
                idx_s = decade_index(rs[k], NINT(log10(rs[k])), o->nis2, ntb_s);
            else
                idx_s = 1;

            if (rg[k] > o->r_g[1]) {
                idx_g = decade_index(rg[k], NINT(log10(rg[k])), o->nig2, ntb_g);
                lamg = 1. / ilamg[k];
                lam_exp = lamg * pow(cgg[3] * ogg2 * ogg1, bm_g);
                N0_exp = ogg1 * rg[k] / am_g * pow(lam_exp, cge[1]);
                idx_g1 = decade_index_d(N0_exp, NINT(log10(N0_exp)), o->nig3, ntb_g1);
            } else {
                idx_g = 1;
                idx_g1 = ntb_g1;
            }

            /* S&C prefactor M:1884-1900 */
            otemp = 1. / temp[k];
            rvs = rho[k] * qvsi[k];
            rvs_p = rvs * otemp * (lsub * otemp * oRv - 1.);
            rvs_pp = rvs * (otemp * (lsub * otemp * oRv - 1.)
                          * otemp * (lsub * otemp * oRv - 1.)
                          + (-2. * lsub * otemp * otemp * otemp * oRv)
                          + otemp * otemp);
            gamsc = lsub * diffu[k] / tcond[k] * rvs_p;
            alphsc = 0.5 * (gamsc / (1. + gamsc)) * (gamsc / (1. + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
            alphsc = MAXD(1.E-9, alphsc);
            xsat = ssati[k];
            if (fabs(xsat) < 1.E-9) xsat = 0.;
            t1_subl = 4. * PI * (1.0 - alphsc * xsat
                               + 2. * alphsc * alphsc * xsat * xsat
                               - 5. * alphsc * alphsc * alphsc * xsat * xsat * xsat)
                    / (1. + gamsc);

            /* riming M:1903-1935 */
            if (L_qc[k] && mvd_c[k] > D0c) {
                xDs = 0.0;
                if (L_qs[k]) xDs = smoc[k] / smob[k];
                if (xDs > D0s) {
                    idx = 1 + (int)(nbs * log(xDs / o->Ds[1]) / log(o->Ds[nbs] / o->Ds[1]));
                    idx = idx < nbs ? idx : nbs;
                    if (idx < 1) idx = 1;
                    Ef_sw = EFSW(o->t_Efsw, idx, clampi((int)(mvd_c[k] * 1.E6), 1, nbc));
                    prs_scw[k] = rhof[k] * o->t1_qs_qc * Ef_sw * rc[k] * smoe[k];
                    pnc_scw[k] = rhof[k] * o->t1_qs_qc * Ef_sw * nc[k] * smoe[k];
                    pnc_scw[k] = MIND(nc[k] * odts, pnc_scw[k]);
                }

                if (rg[k] >= o->r_g[1] && mvd_c[k] > D0c) {
                    xDg = (bm_g + mu_g + 1.) * ilamg[k];
                    vtg = rhof[k] * av_g * cgg[6] * ogg3 * pow(ilamg[k], bv_g);
                    stoke_g = mvd_c[k] * mvd_c[k] * vtg * rho_w / (9. * visco[k] * xDg);
                    if (xDg > D0g) {
                        if (stoke_g >= 0.4 && stoke_g <= 10.)
                            Ef_gw = 0.55 * log10(2.51 * stoke_g);
                        else if (stoke_g < 0.4)
                            Ef_gw = 0.0;
                        else if (stoke_g > 10)
                            Ef_gw = 0.77;
                        prg_gcw[k] = rhof[k] * o->t1_qg_qc * Ef_gw * rc[k] * N0_g[k] * pow(ilamg[k], cge[9]);
                        pnc_gcw[k] = rhof[k] * o->t1_qg_qc * Ef_gw * nc[k] * N0_g[k] * pow(ilamg[k], cge[9]);
                        pnc_gcw[k] = MIND(nc[k] * odts, pnc_gcw[k]);
                    }
                }
            }

            /* aerosol scavenging by snow/graupel: dead outputs, M:1938-1959 */
            if (rs[k] > o->r_s[1]) {
                xDs = smoc[k] / smob[k];
                Ef_sa = Eff_aero(xDs, 0.04E-6, visco[k], rho[k], temp[k], 's');
                pna_sca[k] = rhof[k] * o->t1_qs_qc * Ef_sa * nwfa[k] * smoe[k];
                pna_sca[k] = MIND(nwfa[k] * odts, pna_sca[k]);
                Ef_sa = Eff_aero(xDs, 0.8E-6, visco[k], rho[k], temp[k], 's');
                pnd_scd[k] = rhof[k] * o->t1_qs_qc * Ef_sa * nifa[k] * smoe[k];
                pnd_scd[k] = MIND(nifa[k] * odts, pnd_scd[k]);
            }
            if (rg[k] > o->r_g[1]) {
                xDg = (bm_g + mu_g + 1.) * ilamg[k];
                Ef_ga = Eff_aero(xDg, 0.04E-6, visco[k], rho[k], temp[k], 'g');
                pna_gca[k] = rhof[k] * o->t1_qg_qc * Ef_ga * nwfa[k] * N0_g[k] * pow(ilamg[k], cge[9]);
                pna_gca[k] = MIND(nwfa[k] * odts, pna_gca[k]);
                Ef_ga = Eff_aero(xDg, 0.8E-6, visco[k], rho[k], temp[k], 'g');
                pnd_gcd[k] = rhof[k] * o->t1_qg_qc * Ef_ga * nifa[k] * N0_g[k] * pow(ilamg[k], cge[9]);
                pnd_gcd[k] = MIND(nifa[k] * odts, pnd_gcd[k]);
            }

            /* rain-snow, rain-graupel collection M:1964-2019 */
            if (rr[k] >= o->r_r[1]) {
                if (rs[k] >= o->r_s[1]) {
#define TS(t) RACS(o->t, idx_s, idx_t, idx_r1, idx_r)
                    if (temp[k] < T_0) {
                        prr_rcs[k] = -(TS(tmr_racs2) + TS(tcr_sacr2) + TS(tmr_racs1) + TS(tcr_sacr1));
                        prs_rcs[k] = TS(tmr_racs2) + TS(tcr_sacr2) - TS(tcs_racs1) - TS(tms_sacr1);
                        prg_rcs[k] = TS(tmr_racs1) + TS(tcr_sacr1) + TS(tcs_racs1) + TS(tms_sacr1);
                        prr_rcs[k] = MAXD(-rr[k] * odts, prr_rcs[k]);
                        prs_rcs[k] = MAXD(-rs[k] * odts, prs_rcs[k]);
                        prg_rcs[k] = MIND((rr[k] + rs[k]) * odts, prg_rcs[k]);
                        pnr_rcs[k] = TS(tnr_racs1) + TS(tnr_racs2) + TS(tnr_sacr1) + TS(tnr_sacr2);
                    } else {
                        prs_rcs[k] = -TS(tcs_racs1) - TS(tms_sacr1) + TS(tmr_racs2) + TS(tcr_sacr2);
                        prs_rcs[k] = MAXD(-rs[k] * odts, prs_rcs[k]);
                        prr_rcs[k] = -prs_rcs[k];
                        pnr_rcs[k] = TS(tnr_racs2) + TS(tnr_sacr2);
                    }
#undef TS
                    pnr_rcs[k] = MIND(nr[k] * odts, pnr_rcs[k]);
                }

                if (rg[k] >= o->r_g[1]) {
#define TG(t) RACG(o->t, idx_g1, idx_g, idx_r1, idx_r)
                    if (temp[k] < T_0) {
                        prg_rcg[k] = TG(tmr_racg) + TG(tcr_gacr);
                        prg_rcg[k] = MIND(rr[k] * odts, prg_rcg[k]);
                        prr_rcg[k] = -prg_rcg[k];
                        pnr_rcg[k] = TG(tnr_racg) + TG(tnr_gacr);
                        pnr_rcg[k] = MIND(nr[k] * odts, pnr_rcg[k]);
                    } else {
                        prr_rcg[k] = TG(tcg_racg);
                        prr_rcg[k] = MIND(rg[k] * odts, prr_rcg[k]);
                        prg_rcg[k] = -prr_rcg[k];
                        pnr_rcg[k] = -5. * TG(tnr_gacr);
                    }
#undef TG
                }
            }

            /* ---- below 0C, M:2025-2231 ---- */
            if (temp[k] < T_0) {
                vts_boost[k] = 1.0;
                rate_max = (qv[k] - qvsi[k]) * rho[k] * odts * 0.999;

                /* xni=1000 / idx_IN (M:2043-2062) index no table: dead */

                if (rr[k] > o->r_r[1]) {
                    prg_rfz[k] = QRFZ(o->tpg_qrfz, idx_r, idx_r1, idx_tc) * odts;
                    pri_rfz[k] = QRFZ(o->tpi_qrfz, idx_r, idx_r1, idx_tc) * odts;
                    pni_rfz[k] = QRFZ(o->tni_qrfz, idx_r, idx_r1, idx_tc) * odts;
                    pnr_rfz[k] = QRFZ(o->tnr_qrfz, idx_r, idx_r1, idx_tc) * odts;
                    pnr_rfz[k] = MIND(nr[k] * odts, pnr_rfz[k]);
                } else if (rr[k] > R1 && temp[k] < HGFR) {
                    pri_rfz[k] = rr[k] * odts;
                    pnr_rfz[k] = nr[k] * odts;
                    pni_rfz[k] = pnr_rfz[k];
                }
                if (rc[k] > o->r_c[1]) {
                    pri_wfz[k] = QCFZ(o->tpi_qcfz, idx_c, idx_tc) * odts;
                    pri_wfz[k] = MIND(rc[k] * odts, pri_wfz[k]);
                    pni_wfz[k] = QCFZ(o->tni_qcfz, idx_c, idx_tc) * odts;
                    pni_wfz[k] = MIND(MIND(Nt_c * odts, pri_wfz[k] / (2. * xm0i)), pni_wfz[k]);
                } else if (rc[k] > R1 && temp[k] < HGFR) {
                    pri_wfz[k] = rc[k] * odts;
                    pni_wfz[k] = nc[k] * odts;
                }

                /* Cooper nucleation, or DeMott's dust nucleation when aerosol-aware (dustyIce = .true., M:30), M:2090-2101 */
                if ((ssati[k] >= 0.25) || (ssatw[k] > eps && temp[k] < 253.15)) {
                    if (aero) xnc = iceDeMott(tempc, qv[k], qvs[k], qvsi[k], rho[k], nifa[k]);
                    else      xnc = MIND(250.E3, TNO * exp(ATO * (T_0 - temp[k])));
                    xni = ni[k] + (pni_rfz[k] + pni_wfz[k]) * dtsave;
                    pni_inu[k] = 0.5 * (xnc - xni + fabs(xnc - xni)) * odts;
                    pri_inu[k] = MIND(rate_max, xm0i * pni_inu[k]);
                    pni_inu[k] = pri_inu[k] / xm0i;
                }
                /* freezing of aqueous aerosols, Koop et al. (2001), M:2103-2111 (homogIce = .true., M:31) */
                xni = smo0[k] + ni[k] + (pni_rfz[k] + pni_wfz[k] + pni_inu[k]) * dtsave;
                if (aero && (xni <= 500.E3) && (temp[k] < 238) && (ssati[k] >= 0.4)) {
                    xnc = iceKoop(temp[k], qv[k], qvs[k], nwfa[k], dtsave);
                    pni_iha[k] = xnc * odts;
                    pri_iha[k] = MIND(rate_max, xm0i * 0.1 * pni_iha[k]);
                    pni_iha[k] = pri_iha[k] / (xm0i * 0.1);
                }

                /* ice deposition/sublimation M:2116-2149 */
                if (L_qi[k]) {
                    lami = pow(am_i * cig[2] * oig1 * ni[k] / ri[k], obmi);
                    ilami = 1. / lami;
                    xDi = MAXD(D0i, (bm_i + mu_i + 1.) * ilami);
                    xmi = am_i * pow(xDi, bm_i);
                    oxmi = 1. / xmi;
                    pri_ide[k] = C_cube * t1_subl * diffu[k] * ssati[k] * rvs * oig1 * cig[5] * ni[k] * ilami;

                    if (pri_ide[k] < 0.0) {
                        pri_ide[k] = MAXD(MAXD(-ri[k] * odts, pri_ide[k]), rate_max);
                        pni_ide[k] = pri_ide[k] * oxmi;
                        pni_ide[k] = MAXD(-ni[k] * odts, pni_ide[k]);
                    } else {
                        pri_ide[k] = MIND(pri_ide[k], rate_max);
                        prs_ide[k] = (1.0 - IAUS(o->tpi_ide, idx_i, idx_i1)) * pri_ide[k];
                        pri_ide[k] = IAUS(o->tpi_ide, idx_i, idx_i1) * pri_ide[k];
                    }

                    if ((idx_i == ntb_i) || (xDi > 5.0 * D0s)) {
                        prs_iau[k] = ri[k] * .99 * odts;
                        pni_iau[k] = ni[k] * .95 * odts;
                    } else if (xDi < 0.1 * D0s) {
                        prs_iau[k] = 0.;
                        pni_iau[k] = 0.;
                    } else {
                        prs_iau[k] = IAUS(o->tps_iaus, idx_i, idx_i1) * odts;
                        prs_iau[k] = MIND(ri[k] * .99 * odts, prs_iau[k]);
                        pni_iau[k] = IAUS(o->tni_iaus, idx_i, idx_i1) * odts;
                        pni_iau[k] = MIND(ni[k] * .95 * odts, pni_iau[k]);
                    }
                }

                /* snow/graupel deposition M:2153-2175 */
                if (L_qs[k]) {
                    C_snow = C_sqrd + (tempc + 1.5) * (C_cube - C_sqrd) / (-30. + 1.5);
                    C_snow = MAXD(C_sqrd, MIND(C_snow, C_cube));
                    prs_sde[k] = C_snow * t1_subl * diffu[k] * ssati[k] * rvs
                               * (o->t1_qs_sd * smo1[k] + o->t2_qs_sd * rhof2[k] * vsc2[k] * smof[k]);
                    if (prs_sde[k] < 0.)
                        prs_sde[k] = MAXD(MAXD(-rs[k] * odts, prs_sde[k]), rate_max);
                    else
                        prs_sde[k] = MIND(prs_sde[k], rate_max);
                }

                if (L_qg[k] && ssati[k] < -eps) {
                    prg_gde[k] = C_cube * t1_subl * diffu[k] * ssati[k] * rvs
                               * N0_g[k] * (o->t1_qg_sd * pow(ilamg[k], cge[10])
                                          + o->t2_qg_sd * vsc2[k] * rhof2[k] * pow(ilamg[k], cge[11]));
                    if (prg_gde[k] < 0.)
                        prg_gde[k] = MAXD(MAXD(-rg[k] * odts, prg_gde[k]), rate_max);
                    else
                        prg_gde[k] = MIND(prg_gde[k], rate_max);
                }

                /* snow/rain collecting ice M:2178-2202 */
                if (L_qi[k]) {
                    lami = pow(am_i * cig[2] * oig1 * ni[k] / ri[k], obmi);
                    ilami = 1. / lami;
                    xDi = MAXD(D0i, (bm_i + mu_i + 1.) * ilami);
                    xmi = am_i * pow(xDi, bm_i);
                    oxmi = 1. / xmi;
                    if (rs[k] >= o->r_s[1]) {
                        prs_sci[k] = o->t1_qs_qi * rhof[k] * Ef_si * ri[k] * smoe[k];
                        pni_sci[k] = prs_sci[k] * oxmi;
                    }
                    if (rr[k] >= o->r_r[1] && mvd_r[k] > 4. * xDi) {
                        lamr = 1. / ilamr[k];
                        pri_rci[k] = rhof[k] * o->t1_qr_qi * Ef_ri * ri[k] * N0_r[k] * pow(lamr + fv_r, -cre[9]);
                        pnr_rci[k] = rhof[k] * o->t1_qr_qi * Ef_ri * ni[k] * N0_r[k] * pow(lamr + fv_r, -cre[9]);
                        pni_rci[k] = pri_rci[k] * oxmi;
                        prr_rci[k] = rhof[k] * o->t2_qr_qi * Ef_ri * ni[k] * N0_r[k] * pow(lamr + fv_r, -cre[8]);
                        prr_rci[k] = MIND(rr[k] * odts, prr_rci[k]);
                        prg_rci[k] = pri_rci[k] + prr_rci[k];
                    }
                }

                /* Hallett-Mossop M:2205-2218 */
                if (prg_gcw[k] > eps && tempc > -8.0) {
                    tf = 0.;
                    if (tempc >= -5.0 && tempc < -3.0)
                        tf = 0.5 * (-3.0 - tempc);
                    else if (tempc > -8.0 && tempc < -5.0)
                        tf = 0.33333333 * (8.0 + tempc);
                    pni_ihm[k] = 3.5E8 * tf * prg_gcw[k];
                    pri_ihm[k] = xm0i * pni_ihm[k];
                    prs_ihm[k] = prs_scw[k] / (prs_scw[k] + prg_gcw[k]) * pri_ihm[k];
                    prg_ihm[k] = prg_gcw[k] / (prs_scw[k] + prg_gcw[k]) * pri_ihm[k];
                }

                /* rimed snow -> graupel M:2224-2231 */
                if (prs_scw[k] > 2.0 * prs_sde[k] && prs_sde[k] > eps) {
                    r_frac = MIND(30.0, prs_scw[k] / prs_sde[k]);
                    g_frac = MIND(0.95, 0.15 + (r_frac - 2.) * .028);
                    vts_boost[k] = MIND(1.5, 1.1 + (r_frac - 2.) * .016);
                    prg_scw[k] = g_frac * prs_scw[k];
                    prs_scw[k] = (1. - g_frac) * prs_scw[k];
                }
            } else {
                /* ---- at/above 0C: melting, M:2237-2281 ---- */
                if (L_qs[k]) {
                    prr_sml[k] = (tempc * tcond[k] - lvap0 * diffu[k] * delQvs[k])
                               * (o->t1_qs_me * smo1[k] + o->t2_qs_me * rhof2[k] * vsc2[k] * smof[k]);
                    prr_sml[k] = prr_sml[k] + 4218. * olfus * tempc * (prr_rcs[k] + prs_scw[k]);
                    prr_sml[k] = MIND(rs[k] * odts, MAXD(0., prr_sml[k]));
                    pnr_sml[k] = smo0[k] / rs[k] * prr_sml[k] * pow(10.0, -0.25 * tempc);
                    pnr_sml[k] = MIND(smo0[k] * odts, pnr_sml[k]);
                    if (ssati[k] < 0.) {
                        prs_sde[k] = C_cube * t1_subl * diffu[k] * ssati[k] * rvs
                                   * (o->t1_qs_sd * smo1[k] + o->t2_qs_sd * rhof2[k] * vsc2[k] * smof[k]);
                        prs_sde[k] = MAXD(-rs[k] * odts, prs_sde[k]);
                    }
                }

                if (L_qg[k]) {
                    prr_gml[k] = (tempc * tcond[k] - lvap0 * diffu[k] * delQvs[k])
                               * N0_g[k] * (o->t1_qg_me * pow(ilamg[k], cge[10])
                                          + o->t2_qg_me * rhof2[k] * vsc2[k] * pow(ilamg[k], cge[11]));
                    prr_gml[k] = MIND(rg[k] * odts, MAXD(0., prr_gml[k]));
                    pnr_gml[k] = N0_g[k] * cgg[2] * pow(ilamg[k], cge[2]) / rg[k]
                               * prr_gml[k] * pow(10.0, -0.5 * tempc);
                    if (ssati[k] < 0.) {
                        prg_gde[k] = C_cube * t1_subl * diffu[k] * ssati[k] * rvs
                                   * N0_g[k] * (o->t1_qg_sd * pow(ilamg[k], cge[10])
                                              + o->t2_qg_sd * vsc2[k] * rhof2[k] * pow(ilamg[k], cge[11]));
                        prg_gde[k] = MAXD(-rg[k] * odts, prg_gde[k]);
                    }
                }
                if (dt > 120.) {                             /* M:2277-2281 */
                    prr_rcw[k] = prr_rcw[k] + prs_scw[k] + prg_gcw[k];
                    prs_scw[k] = 0.;
                    prg_gcw[k] = 0.;
                }
            }
        }
    }

    /* ---- I: conservation limiters, M:2291-2387 ---- */
    for (k = kts; k <= kte; k++) {
        sump = pri_inu[k] + pri_ide[k] + prs_ide[k] + prs_sde[k] + prg_gde[k] + pri_iha[k];
        rate_max = (qv[k] - qvsi[k]) * odts * 0.999;
        if ((sump > eps && sump > rate_max) || (sump < -eps && sump < rate_max)) {
            ratio = rate_max / sump;
            pri_inu[k] = pri_inu[k] * ratio;
            pri_ide[k] = pri_ide[k] * ratio;
            pni_ide[k] = pni_ide[k] * ratio;
            prs_ide[k] = prs_ide[k] * ratio;
            prs_sde[k] = prs_sde[k] * ratio;
            prg_gde[k] = prg_gde[k] * ratio;
            pri_iha[k] = pri_iha[k] * ratio;
        }

        sump = -prr_wau[k] - pri_wfz[k] - prr_rcw[k] - prs_scw[k] - prg_scw[k] - prg_gcw[k];
        rate_max = -rc[k] * odts;
        if (sump < rate_max && L_qc[k]) {
            ratio = rate_max / sump;
            prr_wau[k] = prr_wau[k] * ratio;
            pri_wfz[k] = pri_wfz[k] * ratio;
            prr_rcw[k] = prr_rcw[k] * ratio;
            prs_scw[k] = prs_scw[k] * ratio;
            prg_scw[k] = prg_scw[k] * ratio;
            prg_gcw[k] = prg_gcw[k] * ratio;
        }

        sump = pri_ide[k] - prs_iau[k] - prs_sci[k] - pri_rci[k];
        rate_max = -ri[k] * odts;
        if (sump < rate_max && L_qi[k]) {
            ratio = rate_max / sump;
            pri_ide[k] = pri_ide[k] * ratio;
            prs_iau[k] = prs_iau[k] * ratio;
            prs_sci[k] = prs_sci[k] * ratio;
            pri_rci[k] = pri_rci[k] * ratio;
        }

        sump = -prg_rfz[k] - pri_rfz[k] - prr_rci[k] + prr_rcs[k] + prr_rcg[k];
        rate_max = -rr[k] * odts;
        if (sump < rate_max && L_qr[k]) {
            ratio = rate_max / sump;
            prg_rfz[k] = prg_rfz[k] * ratio;
            pri_rfz[k] = pri_rfz[k] * ratio;
            prr_rci[k] = prr_rci[k] * ratio;
            prr_rcs[k] = prr_rcs[k] * ratio;
            prr_rcg[k] = prr_rcg[k] * ratio;
        }

        sump = prs_sde[k] - prs_ihm[k] - prr_sml[k] + prs_rcs[k];
        rate_max = -rs[k] * odts;
        if (sump < rate_max && L_qs[k]) {
            ratio = rate_max / sump;
            prs_sde[k] = prs_sde[k] * ratio;
            prs_ihm[k] = prs_ihm[k] * ratio;
            prr_sml[k] = prr_sml[k] * ratio;
            prs_rcs[k] = prs_rcs[k] * ratio;
        }

        sump = prg_gde[k] - prg_ihm[k] - prr_gml[k] + prg_rcg[k];
        rate_max = -rg[k] * odts;
        if (sump < rate_max && L_qg[k]) {
            ratio = rate_max / sump;
            prg_gde[k] = prg_gde[k] * ratio;
            prg_ihm[k] = prg_ihm[k] * ratio;
            prr_gml[k] = prr_gml[k] * ratio;
            prg_rcg[k] = prg_rcg[k] * ratio;
        }

        pri_ihm[k] = prs_ihm[k] + prg_ihm[k];
        ratio = MIND(fabs(prr_rcg[k]), fabs(prg_rcg[k]));
        prr_rcg[k] = ratio * copysign(1.0, prr_rcg[k]);       /* SIGN(1.0,SNGL(x)), M:2379 */
        prg_rcg[k] = -prr_rcg[k];
        if (temp[k] > T_0) {
            ratio = MIND(fabs(prr_rcs[k]), fabs(prs_rcs[k]));
            prr_rcs[k] = ratio * copysign(1.0, prr_rcs[k]);
            prs_rcs[k] = -prr_rcs[k];
        }
    }

    /* ---- J: tendencies + number re-balance, M:2393-2569 ---- */
    for (k = kts; k <= kte; k++) {
        orho = 1. / rho[k];
        lfus2 = lsub - lvap[k];

        if (aero) {                                          /* M:2397-2408 (dustyIce = .true.) */
            nwfaten[k] = nwfaten[k] - (pna_rca[k] + pna_sca[k] + pna_gca[k] + pni_iha[k]) * orho;
            nifaten[k] = nifaten[k] - (pnd_rcd[k] + pnd_scd[k] + pnd_gcd[k]) * orho;
            nifaten[k] = nifaten[k] - pni_inu[k] * orho;
        }

        qvten[k] = qvten[k] + (-pri_inu[k] - pri_iha[k] - pri_ide[k] - prs_ide[k] - prs_sde[k] - prg_gde[k]) * orho;

        qcten[k] = qcten[k] + (-prr_wau[k] - pri_wfz[k] - prr_rcw[k] - prs_scw[k] - prg_scw[k] - prg_gcw[k]) * orho;

        ncten[k] = ncten[k] + (-pnc_wau[k] - pnc_rcw[k] - pni_wfz[k] - pnc_scw[k] - pnc_gcw[k]) * orho;

        xrc = MAXD(R1, (qc1d[k] + qcten[k] * dtsave) * rho[k]);
        xnc = MAXD(2., (nc1d[k] + ncten[k] * dtsave) * rho[k]);
        if (xrc > R1) {
            nu_c = NINT(1000.E6 / xnc) + 2; if (nu_c > 15) nu_c = 15;
            lamc = pow(xnc * am_r * ccg2[nu_c] * ocg1[nu_c] / rc[k], obmr);
            xDc = (bm_r + nu_c + 1.) / lamc;
            if (xDc < D0c) {
                lamc = cce2[nu_c] / D0c;
                xnc = ccg1[nu_c] * ocg2[nu_c] * xrc / am_r * pow(lamc, bm_r);
                ncten[k] = (xnc - nc1d[k] * rho[k]) * odts * orho;
            } else if (xDc > D0r * 2.) {
                lamc = cce2[nu_c] / (D0r * 2.);
                xnc = ccg1[nu_c] * ocg2[nu_c] * xrc / am_r * pow(lamc, bm_r);
                ncten[k] = (xnc - nc1d[k] * rho[k]) * odts * orho;
            }
        } else {
            ncten[k] = -nc1d[k] * odts;
        }
        xnc = MAXD(0., (nc1d[k] + ncten[k] * dtsave) * rho[k]);
        if (xnc > Nt_c_max)
            ncten[k] = (Nt_c_max - nc1d[k] * rho[k]) * odts * orho;

        qiten[k] = qiten[k] + (pri_inu[k] + pri_iha[k] + pri_ihm[k] + pri_wfz[k] + pri_rfz[k] + pri_ide[k]
                             - prs_iau[k] - prs_sci[k] - pri_rci[k]) * orho;

        niten[k] = niten[k] + (pni_inu[k] + pni_iha[k] + pni_ihm[k] + pni_wfz[k] + pni_rfz[k] + pni_ide[k]
                             - pni_iau[k] - pni_sci[k] - pni_rci[k]) * orho;

        xri = MAXD(R1, (qi1d[k] + qiten[k] * dtsave) * rho[k]);
        xni = MAXD(R2, (ni1d[k] + niten[k] * dtsave) * rho[k]);
        if (xri > R1) {
            lami = pow(am_i * cig[2] * oig1 * xni / xri, obmi);
            ilami = 1. / lami;
            xDi = (bm_i + mu_i + 1.) * ilami;
            if (xDi < 5.E-6) {
                lami = cie[2] / 5.E-6;
                xni = MIND(499.e3, cig[1] * oig2 * xri / am_i * pow(lami, bm_i));
                niten[k] = (xni - ni1d[k] * rho[k]) * odts * orho;
            } else if (xDi > 300.E-6) {
                lami = cie[2] / 300.E-6;
                xni = cig[1] * oig2 * xri / am_i * pow(lami, bm_i);
                niten[k] = (xni - ni1d[k] * rho[k]) * odts * orho;
            }
        } else {
            niten[k] = -ni1d[k] * odts;
        }
        xni = MAXD(0., (ni1d[k] + niten[k] * dtsave) * rho[k]);
        if (xni > 499.E3)
            niten[k] = (499.E3 - ni1d[k] * rho[k]) * odts * orho;

        qrten[k] = qrten[k] + (prr_wau[k] + prr_rcw[k] + prr_sml[k] + prr_gml[k] + prr_rcs[k]
                             + prr_rcg[k] - prg_rfz[k] - pri_rfz[k] - prr_rci[k]) * orho;

        nrten[k] = nrten[k] + (pnr_wau[k] + pnr_sml[k] + pnr_gml[k]
                             - (pnr_rfz[k] + pnr_rcr[k] + pnr_rcg[k] + pnr_rcs[k] + pnr_rci[k])) * orho;

        xrr = MAXD(R1, (qr1d[k] + qrten[k] * dtsave) * rho[k]);
        xnr = MAXD(R2, (nr1d[k] + nrten[k] * dtsave) * rho[k]);
        if (xrr > R1) {
            lamr = pow(am_r * crg[3] * org2 * xnr / xrr, obmr);
            mvd_r[k] = (3.0 + mu_r + 0.672) / lamr;
            if (mvd_r[k] > 2.5E-3) {
                mvd_r[k] = 2.5E-3;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                xnr = crg[2] * org3 * xrr * pow(lamr, bm_r) / am_r;
                nrten[k] = (xnr - nr1d[k] * rho[k]) * odts * orho;
            } else if (mvd_r[k] < D0r * 0.75) {
                mvd_r[k] = D0r * 0.75;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                xnr = crg[2] * org3 * xrr * pow(lamr, bm_r) / am_r;
                nrten[k] = (xnr - nr1d[k] * rho[k]) * odts * orho;
            }
        } else {
            qrten[k] = -qr1d[k] * odts;
            nrten[k] = -nr1d[k] * odts;
        }

        qsten[k] = qsten[k] + (prs_iau[k] + prs_sde[k] + prs_sci[k] + prs_scw[k] + prs_rcs[k]
                             + prs_ide[k] - prs_ihm[k] - prr_sml[k]) * orho;

        qgten[k] = qgten[k] + (prg_scw[k] + prg_rfz[k] + prg_gde[k] + prg_rcg[k] + prg_gcw[k]
                             + prg_rci[k] + prg_rcs[k] - prg_ihm[k] - prr_gml[k]) * orho;

        if (temp[k] < T_0) {
            tten[k] = tten[k]
                    + (lsub * ocp[k] * (pri_inu[k] + pri_ide[k] + prs_ide[k] + prs_sde[k] + prg_gde[k] + pri_iha[k])
                     + lfus2 * ocp[k] * (pri_wfz[k] + pri_rfz[k] + prg_rfz[k] + prs_scw[k] + prg_scw[k] + prg_gcw[k]
                                       + prg_rcs[k] + prs_rcs[k] + prr_rci[k] + prg_rcg[k])
                      ) * orho * (1 - IFDRY);
        } else {
            tten[k] = tten[k]
                    + (lfus * ocp[k] * (-prr_sml[k] - prr_gml[k] - prr_rcg[k] - prr_rcs[k])
                     + lsub * ocp[k] * (prs_sde[k] + prg_gde[k])
                      ) * orho * (1 - IFDRY);
        }
    }

    /* ---- K: update for TAU+1 before condensation & sedimentation, M:2574-2656 ---- */
    for (k = kts; k <= kte; k++) {
        temp[k] = t1d[k] + DT * tten[k];
        otemp = 1. / temp[k];
        tempc = temp[k] - 273.15;
        qv[k] = MAXD(1.E-10, qv1d[k] + DT * qvten[k]);
        rho[k] = 0.622 * pres[k] / (R_gas * temp[k] * (qv[k] + 0.622));
        rhof[k] = sqrt(rho_not / rho[k]);
        rhof2[k] = sqrt(rhof[k]);
        qvs[k] = rslf(pres[k], temp[k]);
        ssatw[k] = qv[k] / qvs[k] - 1.;
        if (fabs(ssatw[k]) < eps) ssatw[k] = 0.0;
        diffu[k] = 2.11E-5 * pow(temp[k] / 273.15, 1.94) * (101325. / pres[k]);
        if (tempc >= 0.0)
            visco[k] = (1.718 + 0.0049 * tempc) * 1.0E-5;
        else
            visco[k] = (1.718 + 0.0049 * tempc - 1.2E-5 * tempc * tempc) * 1.0E-5;
        vsc2[k] = sqrt(rho[k] / visco[k]);
        lvap[k] = lvap0 + (2106.0 - 4218.0) * tempc;
        tcond[k] = (5.69 + 0.0168 * tempc) * 1.0E-5 * 418.936;
        ocp[k] = 1. / (Cp * (1. + 0.887 * qv[k]));
        lvt2[k] = lvap[k] * lvap[k] * ocp[k] * oRv * otemp * otemp;

        nwfa[k] = MAXD(11.1E6, (nwfa1d[k] + nwfaten[k] * DT) * rho[k]);

        if ((qc1d[k] + qcten[k] * DT) > R1) {
            rc[k] = (qc1d[k] + qcten[k] * DT) * rho[k];
            nc[k] = MAXD(2., (nc1d[k] + ncten[k] * DT) * rho[k]);
            if (!aero) nc[k] = Nt_c;                         /* M:2602 */
            L_qc[k] = 1;
        } else {
            rc[k] = R1;
            nc[k] = 2.;
            L_qc[k] = 0;
        }

        if ((qi1d[k] + qiten[k] * DT) > R1) {
            ri[k] = (qi1d[k] + qiten[k] * DT) * rho[k];
            ni[k] = MAXD(R2, (ni1d[k] + niten[k] * DT) * rho[k]);
            L_qi[k] = 1;
        } else {
            ri[k] = R1;
            ni[k] = R2;
            L_qi[k] = 0;
        }

        if ((qr1d[k] + qrten[k] * DT) > R1) {
            rr[k] = (qr1d[k] + qrten[k] * DT) * rho[k];
            nr[k] = MAXD(R2, (nr1d[k] + nrten[k] * DT) * rho[k]);
            L_qr[k] = 1;
            lamr = pow(am_r * crg[3] * org2 * nr[k] / rr[k], obmr);
            mvd_r[k] = (3.0 + mu_r + 0.672) / lamr;
            if (mvd_r[k] > 2.5E-3) {
                mvd_r[k] = 2.5E-3;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                nr[k] = crg[2] * org3 * rr[k] * pow(lamr, bm_r) / am_r;
            } else if (mvd_r[k] < D0r * 0.75) {
                mvd_r[k] = D0r * 0.75;
                lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
                nr[k] = crg[2] * org3 * rr[k] * pow(lamr, bm_r) / am_r;
            }
        } else {
            rr[k] = R1;
            nr[k] = R2;
            L_qr[k] = 0;
        }

        if ((qs1d[k] + qsten[k] * DT) > R1) {
            rs[k] = (qs1d[k] + qsten[k] * DT) * rho[k];
            L_qs[k] = 1;
        } else {
            rs[k] = R1;
            L_qs[k] = 0;
        }

        if ((qg1d[k] + qgten[k] * DT) > R1) {
            rg[k] = (qg1d[k] + qgten[k] * DT) * rho[k];
            L_qg[k] = 1;
        } else {
            rg[k] = R1;
            L_qg[k] = 0;
        }
    }

    /* ---- L: PSD refresh, M:2662-2750 ---- */
    if (!iiwarm) {
        for (k = kts; k <= kte; k++) {
            if (!L_qs[k]) continue;
            tc0 = MIND(-0.1, temp[k] - 273.15);
            smob[k] = rs[k] * oams;
            if (bm_s > (2.0 - 1.e-3) && bm_s < (2.0 + 1.e-3)) {
                smo2[k] = smob[k];
            } else {
                loga_ = mom_loga(o->sa, tc0, bm_s);
                a_ = pow(10.0, loga_);
                b_ = mom_b(o->sb, tc0, bm_s);
                smo2[k] = pow(smob[k] / a_, 1. / b_);
            }
            loga_ = mom_loga(o->sa, tc0, cse[1]);
            a_ = pow(10.0, loga_);
            b_ = mom_b(o->sb, tc0, cse[1]);
            smoc[k] = a_ * pow(smo2[k], b_);
            loga_ = mom_loga(o->sa, tc0, cse[14]);
            a_ = pow(10.0, loga_);
            b_ = mom_b(o->sb, tc0, cse[14]);
            smod[k] = a_ * pow(smo2[k], b_);
        }

        N0_min = gonv_max;
        k_0 = kts;
        for (k = kte; k >= kts; k--)
            if (temp[k] >= 270.65) k_0 = k_0 > k ? k_0 : k;
        for (k = kte; k >= kts; k--) {
            if (k > k_0 && L_qr[k] && mvd_r[k] > 100.E-6)
                xslw1 = 4.01 + log10(mvd_r[k]);
            else
                xslw1 = 0.01;
            ygra1 = 4.31 + log10(MAXD(5.E-5, rg[k]));
            zans1 = 3.1 + (100. / (300. * xslw1 * ygra1 / (10. / xslw1 + 1. + 0.25 * ygra1) + 30. + 10. * ygra1));
            N0_exp = pow(10., zans1);
            N0_exp = MAXD(gonv_min, MIND(N0_exp, gonv_max));
            N0_min = MIND(N0_exp, N0_min);
            N0_exp = N0_min;
            lam_exp = pow(N0_exp * am_g * cgg[1] / rg[k], oge1);
            lamg = lam_exp * pow(cgg[3] * ogg2 * ogg1, obmg);
            ilamg[k] = 1. / lamg;
            N0_g[k] = N0_exp / (cgg[2] * lam_exp) * pow(lamg, cge[2]);
        }
    }

    for (k = kte; k >= kts; k--) {                           /* M:2745-2750 */
        lamr = pow(am_r * crg[3] * org2 * nr[k] / rr[k], obmr);
        ilamr[k] = 1. / lamr;
        mvd_r[k] = (3.0 + mu_r + 0.672) / lamr;
        N0_r[k] = nr[k] * org2 * pow(lamr, cre[2]);
    }

    /* ---- M: cloud water condensation/evaporation, M:2780-2874 ---- */
    for (k = kts; k <= kte; k++) {
        orho = 1. / rho[k];
        if ((ssatw[k] > eps) || (ssatw[k] < -eps && L_qc[k])) {
            clap = (qv[k] - qvs[k]) / (1. + lvt2[k] * qvs[k]);
            for (n = 1; n <= 3; n++) {
                fcd = qvs[k] * exp(lvt2[k] * clap) - qv[k] + clap;
                dfcd = qvs[k] * lvt2[k] * exp(lvt2[k] * clap) + 1.;
                clap = clap - fcd / dfcd;
            }
            xrc = rc[k] + clap * rho[k];
            xnc = 0.;
            if (xrc > R1) {
                prw_vcd[k] = clap * odt;
                if (clap > eps) {
                    if (aero) xnc = MAXD((real)2., activ_ncloud(temp[k], w1d[k], nwfa[k]));   /* M:2796-2797 */
                    else      xnc = Nt_c;
                    pnc_wcd[k] = 0.5 * (xnc - nc[k] + fabs(xnc - nc[k])) * odts * orho;
                } else if (clap < -eps && ssatw[k] < -1.E-6 && aero) {      /* droplet evaporation, M:2804-2852 */
                    tempc = temp[k] - 273.15;
                    otemp = 1. / temp[k];
                    rvs = rho[k] * qvs[k];
                    rvs_p = rvs * otemp * (lvap[k] * otemp * oRv - 1.);
                    rvs_pp = rvs * (otemp * (lvap[k] * otemp * oRv - 1.) * otemp * (lvap[k] * otemp * oRv - 1.)
                                    + (-2. * lvap[k] * otemp * otemp * otemp * oRv) + otemp * otemp);
                    gamsc = lvap[k] * diffu[k] / tcond[k] * rvs_p;
                    alphsc = 0.5 * (gamsc / (1. + gamsc)) * (gamsc / (1. + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
                    alphsc = MAXD(1.E-9, alphsc);
                    xsat = ssatw[k];
                    if (fabs(xsat) < 1.E-9) xsat = 0.;
                    t1_evap = 2. * PI * (1.0 - alphsc * xsat + 2. * alphsc * alphsc * xsat * xsat
                                         - 5. * alphsc * alphsc * alphsc * xsat * xsat * xsat) / (1. + gamsc);
                    double Dc_star = sqrt((double)-2. * DT * t1_evap / (2. * PI) * 4. * diffu[k] * ssatw[k] * rvs / rho_w);
                    int idx_d = clampi((int)(1.E6 * Dc_star), 1, nbc);
                    int idx_n = NINT(1.0 + (real)nbc * log(nc[k] / o->t_Nc[1]) / o->nic1);
                    idx_n = clampi(idx_n, 1, nbc);
                    if (rc[k] > o->r_c[1]) idx_c = decade_index(rc[k], NINT(log10(rc[k])), o->nic2, ntb_c);
                    else idx_c = 1;
                    prw_vcd[k] = MAXD((double)(-rc[k] * 0.99 * orho * odt), prw_vcd[k]);
                    pnc_wcd[k] = MAXD((double)(-nc[k] * 0.99 * orho * odt),
                                      (double)(-o->tnc_wev[IX3(idx_d, idx_c, idx_n, nbc, ntb_c)] * orho * odt));
                }
            } else {
                prw_vcd[k] = -rc[k] * orho * odt;
                pnc_wcd[k] = -nc[k] * orho * odt;
            }

            qvten[k] = qvten[k] - prw_vcd[k];
            qcten[k] = qcten[k] + prw_vcd[k];
            ncten[k] = ncten[k] + pnc_wcd[k];
            nwfaten[k] = nwfaten[k] - pnc_wcd[k];
            tten[k] = tten[k] + lvap[k] * ocp[k] * prw_vcd[k] * (1 - IFDRY);
            rc[k] = MAXD(R1, (qc1d[k] + DT * qcten[k]) * rho[k]);
            nc[k] = MAXD(2., (nc1d[k] + DT * ncten[k]) * rho[k]);
            if (!aero) nc[k] = Nt_c;                         /* M:2867 */
            qv[k] = MAXD(1.E-10, qv1d[k] + DT * qvten[k]);
            temp[k] = t1d[k] + DT * tten[k];
            rho[k] = 0.622 * pres[k] / (R_gas * temp[k] * (qv[k] + 0.622));
            qvs[k] = rslf(pres[k], temp[k]);
            ssatw[k] = qv[k] / qvs[k] - 1.;
        }
    }

    /* ---- N: rain evaporation, M:2880-2960 ---- */
    for (k = kts; k <= kte; k++) {
        if ((ssatw[k] < -eps) && L_qr[k] && (!(prw_vcd[k] > 0.))) {
            tempc = temp[k] - 273.15;
            otemp = 1. / temp[k];
            orho = 1. / rho[k];
            rhof[k] = sqrt(rho_not * orho);
            rhof2[k] = sqrt(rhof[k]);
            diffu[k] = 2.11E-5 * pow(temp[k] / 273.15, 1.94) * (101325. / pres[k]);
            if (tempc >= 0.0)
                visco[k] = (1.718 + 0.0049 * tempc) * 1.0E-5;
            else
                visco[k] = (1.718 + 0.0049 * tempc - 1.2E-5 * tempc * tempc) * 1.0E-5;
            vsc2[k] = sqrt(rho[k] / visco[k]);
            lvap[k] = lvap0 + (2106.0 - 4218.0) * tempc;
            tcond[k] = (5.69 + 0.0168 * tempc) * 1.0E-5 * 418.936;
            ocp[k] = 1. / (Cp * (1. + 0.887 * qv[k]));

            rvs = rho[k] * qvs[k];
            rvs_p = rvs * otemp * (lvap[k] * otemp * oRv - 1.);
            rvs_pp = rvs * (otemp * (lvap[k] * otemp * oRv - 1.)
                          * otemp * (lvap[k] * otemp * oRv - 1.)
                          + (-2. * lvap[k] * otemp * otemp * otemp * oRv)
                          + otemp * otemp);
            gamsc = lvap[k] * diffu[k] / tcond[k] * rvs_p;
            alphsc = 0.5 * (gamsc / (1. + gamsc)) * (gamsc / (1. + gamsc)) * rvs_pp / rvs_p * rvs / rvs_p;
            alphsc = MAXD(1.E-9, alphsc);
            xsat = MIND(-1.E-9, ssatw[k]);
            t1_evap = 2. * PI * (1.0 - alphsc * xsat
                               + 2. * alphsc * alphsc * xsat * xsat
                               - 5. * alphsc * alphsc * alphsc * xsat * xsat * xsat)
                    / (1. + gamsc);

            lamr = 1. / ilamr[k];

            if (qv[k] / qvs[k] < 0.95 && rr[k] * orho <= 1.E-8) {
                prv_rev[k] = rr[k] * orho * odts;
            } else {
                prv_rev[k] = t1_evap * diffu[k] * (-ssatw[k]) * N0_r[k] * rvs
                           * (o->t1_qr_ev * pow(ilamr[k], cre[10])
                            + o->t2_qr_ev * vsc2[k] * rhof2[k] * pow(lamr + 0.5 * fv_r, -cre[11]));
                rate_max = MIND((rr[k] * orho * odts), (qvs[k] - qv[k]) * odts);
                prv_rev[k] = MIND(rate_max, prv_rev[k] * orho);

                if (prr_gml[k] > 0.0) {
                    eva_factor = MIND(1.0, 0.01 + (0.99 - 0.01) * (tempc / 20.0));
                    prv_rev[k] = prv_rev[k] * eva_factor;
                }
            }

            pnr_rev[k] = MIND(nr[k] * 0.99 * orho * odts, prv_rev[k] * nr[k] / rr[k]);

            qrten[k] = qrten[k] - prv_rev[k];
            qvten[k] = qvten[k] + prv_rev[k];
            nrten[k] = nrten[k] - pnr_rev[k];
            nwfaten[k] = nwfaten[k] + pnr_rev[k];
            tten[k] = tten[k] - lvap[k] * ocp[k] * prv_rev[k] * (1 - IFDRY);

            rr[k] = MAXD(R1, (qr1d[k] + DT * qrten[k]) * rho[k]);
            qv[k] = MAXD(1.E-10, qv1d[k] + DT * qvten[k]);
            nr[k] = MAXD(R2, (nr1d[k] + DT * nrten[k]) * rho[k]);
            temp[k] = t1d[k] + DT * tten[k];
            rho[k] = 0.622 * pres[k] / (R_gas * temp[k] * (qv[k] + 0.622));
        }

        /* ---- N': KiD rate diagnostics, save_dg order of M:2967-3119 ---- */
        if (rates) {
            const double *src[TH_ORACLE_NRATES] = {
                pri_inu, pri_ide, prs_ide, prs_sde, prg_gde, pri_wfz, prs_scw, prg_scw, prg_gcw, pri_ihm,
                pri_rfz, prs_iau, prs_sci, pri_rci, pni_inu, pni_ihm, pni_wfz, pni_rfz, pni_ide, pni_iau,
                pni_sci, pni_rci, prr_sml, prr_gml, pnr_rcs, pnr_rcg, pnr_rci, pnr_sml, pnr_gml, pnr_rfz,
                prr_wau, prr_rcw, prv_rev, pnr_wau, pnr_rev, pnr_rcr };
            for (int r = iiwarm ? 30 : 0; r < TH_ORACLE_NRATES; r++)
                rates[(size_t)r * nz + k] = src[r][k];
        }
    }

    /* ---- O: fall speeds and substep counts, M:3206-3354 ---- */
    nstep = 0;
    for (n = 1; n <= 5; n++) { onstep[n] = 1.0; ksed1[n] = 0 /* = kts */; }
    /* vt*k(kts:kte+1) are zero from calloc, M:3209-3216 */
    for (k = kte; k >= kts; k--) {
        vtr = 0.;
        rhof[k] = sqrt(rho_not / rho[k]);

        if (rr[k] > R1) {
            lamr = pow(am_r * crg[3] * org2 * nr[k] / rr[k], obmr);
            vtr = rhof[k] * av_r * crg[6] * org3 * pow(lamr, cre[3]) * pow(lamr + fv_r, -cre[6]);
            vtrk[k] = vtr;
            vtr = rhof[k] * av_r * crg[7] / crg[12] * pow(lamr, cre[12]) * pow(lamr + fv_r, -cre[7]);
            vtnrk[k] = vtr;
        } else {
            vtrk[k] = vtrk[k + 1];
            vtnrk[k] = vtnrk[k + 1];
        }

        if (MAXD(vtrk[k], vtnrk[k]) > 1.E-3) {
            ksed1[1] = ksed1[1] > k ? ksed1[1] : k;
            delta_tp = dzq[k] / (MAXD(vtrk[k], vtnrk[k]));
            int ns = (int)(DT / delta_tp + 1.);
            nstep = nstep > ns ? nstep : ns;
        }
    }
    if (ksed1[1] == kte) ksed1[1] = kte - 1;
    if (nstep > 0) onstep[1] = 1. / (real)nstep;

    if (!iiwarm) {
        nstep = 0;
        for (k = kte; k >= kts; k--) {
            vti = 0.;
            if (ri[k] > R1) {
                lami = pow(am_i * cig[2] * oig1 * ni[k] / ri[k], obmi);
                ilami = 1. / lami;
                vti = rhof[k] * av_i * cig[3] * oig2 * pow(ilami, bv_i);
                vtik[k] = vti;
                vti = rhof[k] * av_i * cig[6] / cig[7] * pow(ilami, bv_i);
                vtnik[k] = vti;
            } else {
                vtik[k] = vtik[k + 1];
                vtnik[k] = vtnik[k + 1];
            }
            if (vtik[k] > 1.E-3) {
                ksed1[2] = ksed1[2] > k ? ksed1[2] : k;
                delta_tp = dzq[k] / vtik[k];
                int ns = (int)(DT / delta_tp + 1.);
                nstep = nstep > ns ? nstep : ns;
            }
        }
        if (ksed1[2] == kte) ksed1[2] = kte - 1;
        if (nstep > 0) onstep[2] = 1. / (real)nstep;

        nstep = 0;
        for (k = kte; k >= kts; k--) {
            vts = 0.;
            if (rs[k] > R1) {
                xDs = smoc[k] / smob[k];
                Mrat = 1. / xDs;
                ils1 = 1. / (Mrat * Lam0 + fv_s);
                ils2 = 1. / (Mrat * Lam1 + fv_s);
                t1_vts = Kap0 * csg[4] * pow(ils1, cse[4]);
                t2_vts = Kap1 * pow(Mrat, mu_s) * csg[10] * pow(ils2, cse[10]);
                ils1 = 1. / (Mrat * Lam0);
                ils2 = 1. / (Mrat * Lam1);
                t3_vts = Kap0 * csg[1] * pow(ils1, cse[1]);
                t4_vts = Kap1 * pow(Mrat, mu_s) * csg[7] * pow(ils2, cse[7]);
                vts = rhof[k] * av_s * (t1_vts + t2_vts) / (t3_vts + t4_vts);
                if (temp[k] > (T_0 + 0.1))
                    vtsk[k] = MAXD(vts * vts_boost[k], vts * ((vtrk[k] - vts * vts_boost[k]) / (temp[k] - T_0)));
                else
                    vtsk[k] = vts * vts_boost[k];
            } else {
                vtsk[k] = vtsk[k + 1];
            }
            if (vtsk[k] > 1.E-3) {
                ksed1[3] = ksed1[3] > k ? ksed1[3] : k;
                delta_tp = dzq[k] / vtsk[k];
                int ns = (int)(DT / delta_tp + 1.);
                nstep = nstep > ns ? nstep : ns;
            }
        }
        if (ksed1[3] == kte) ksed1[3] = kte - 1;
        if (nstep > 0) onstep[3] = 1. / (real)nstep;

        nstep = 0;
        for (k = kte; k >= kts; k--) {
            vtg = 0.;
            if (rg[k] > R1) {
                vtg = rhof[k] * av_g * cgg[6] * ogg3 * pow(ilamg[k], bv_g);
                if (temp[k] > T_0)
                    vtgk[k] = MAXD(vtg, vtrk[k]);
                else
                    vtgk[k] = vtg;
            } else {
                vtgk[k] = vtgk[k + 1];
            }
            if (vtgk[k] > 1.E-3) {
                ksed1[4] = ksed1[4] > k ? ksed1[4] : k;
                delta_tp = dzq[k] / vtgk[k];
                int ns = (int)(DT / delta_tp + 1.);
                nstep = nstep > ns ? nstep : ns;
            }
        }
        if (ksed1[4] == kte) ksed1[4] = kte - 1;
        if (nstep > 0) onstep[4] = 1. / (real)nstep;
    } else {
        for (k = kte; k >= kts; k--) { vtik[k] = 0.; vtnik[k] = 0.; vtsk[k] = 0.; vtgk[k] = 0.; }
    }

    /* ---- P: sedimentation, M:3365-3578 ---- */
    nstep = NINT(1. / onstep[1]);
    if (nstep_out) nstep_out[0] = nstep;
    for (n = 1; n <= nstep; n++) {
        for (k = kte; k >= kts; k--) {
            sed_r[k] = vtrk[k] * rr[k];
            sed_n[k] = vtnrk[k] * nr[k];
        }
        k = kte;
        odzq = 1. / dzq[k];
        orho = 1. / rho[k];
        qrten[k] = qrten[k] - sed_r[k] * odzq * onstep[1] * orho;
        nrten[k] = nrten[k] - sed_n[k] * odzq * onstep[1] * orho;
        rr[k] = MAXD(R1, rr[k] - sed_r[k] * odzq * DT * onstep[1]);
        nr[k] = MAXD(R2, nr[k] - sed_n[k] * odzq * DT * onstep[1]);
        for (k = ksed1[1]; k >= kts; k--) {
            odzq = 1. / dzq[k];
            orho = 1. / rho[k];
            qrten[k] = qrten[k] + (sed_r[k + 1] - sed_r[k]) * odzq * onstep[1] * orho;
            nrten[k] = nrten[k] + (sed_n[k + 1] - sed_n[k]) * odzq * onstep[1] * orho;
            rr[k] = MAXD(R1, rr[k] + (sed_r[k + 1] - sed_r[k]) * odzq * DT * onstep[1]);
            nr[k] = MAXD(R2, nr[k] + (sed_n[k + 1] - sed_n[k]) * odzq * DT * onstep[1]);
        }
        if (rr[kts] > R1 * 10.)
            ppt[0] = ppt[0] + sed_r[kts] * DT * onstep[1];
    }

    /* cloud-water sedimentation M:3414-3425: U1, no-op (see header) */

    nstep = NINT(1. / onstep[2]);
    if (nstep_out) nstep_out[1] = nstep;
    for (n = 1; n <= nstep; n++) {
        if (o->l_sediment) {
            for (k = kte; k >= kts; k--) {
                sed_i[k] = vtik[k] * ri[k];
                sed_n[k] = vtnik[k] * ni[k];
            }
        } else {
            for (k = kts; k <= kte; k++) { sed_i[k] = 0.; sed_n[k] = 0.; }
        }
        k = kte;
        odzq = 1. / dzq[k];
        orho = 1. / rho[k];
        qiten[k] = qiten[k] - sed_i[k] * odzq * onstep[2] * orho;
        niten[k] = niten[k] - sed_n[k] * odzq * onstep[2] * orho;
        ri[k] = MAXD(R1, ri[k] - sed_i[k] * odzq * DT * onstep[2]);
        ni[k] = MAXD(R2, ni[k] - sed_n[k] * odzq * DT * onstep[2]);
        for (k = ksed1[2]; k >= kts; k--) {
            odzq = 1. / dzq[k];
            orho = 1. / rho[k];
            qiten[k] = qiten[k] + (sed_i[k + 1] - sed_i[k]) * odzq * onstep[2] * orho;
            niten[k] = niten[k] + (sed_n[k + 1] - sed_n[k]) * odzq * onstep[2] * orho;
            ri[k] = MAXD(R1, ri[k] + (sed_i[k + 1] - sed_i[k]) * odzq * DT * onstep[2]);
            ni[k] = MAXD(R2, ni[k] + (sed_n[k + 1] - sed_n[k]) * odzq * DT * onstep[2]);
        }
        if (ri[kts] > R1 * 10.)
            ppt[3] = ppt[3] + sed_i[kts] * DT * onstep[2];
    }

    nstep = NINT(1. / onstep[3]);
    if (nstep_out) nstep_out[2] = nstep;
    for (n = 1; n <= nstep; n++) {
        if (o->l_sediment) {
            for (k = kte; k >= kts; k--) sed_s[k] = vtsk[k] * rs[k];
        } else {
            for (k = kts; k <= kte; k++) sed_s[k] = 0.;
        }
        k = kte;
        odzq = 1. / dzq[k];
        orho = 1. / rho[k];
        qsten[k] = qsten[k] - sed_s[k] * odzq * onstep[3] * orho;
        rs[k] = MAXD(R1, rs[k] - sed_s[k] * odzq * DT * onstep[3]);
        for (k = ksed1[3]; k >= kts; k--) {
            odzq = 1. / dzq[k];
            orho = 1. / rho[k];
            qsten[k] = qsten[k] + (sed_s[k + 1] - sed_s[k]) * odzq * onstep[3] * orho;
            rs[k] = MAXD(R1, rs[k] + (sed_s[k + 1] - sed_s[k]) * odzq * DT * onstep[3]);
        }
        if (rs[kts] > R1 * 10.)
            ppt[1] = ppt[1] + sed_s[kts] * DT * onstep[3];
    }

    nstep = NINT(1. / onstep[4]);
    if (nstep_out) nstep_out[3] = nstep;
    for (n = 1; n <= nstep; n++) {
        if (o->l_sediment) {
            for (k = kte; k >= kts; k--) sed_g[k] = vtgk[k] * rg[k];
        } else {
            for (k = kts; k <= kte; k++) sed_g[k] = 0.;
        }
        k = kte;
        odzq = 1. / dzq[k];
        orho = 1. / rho[k];
        qgten[k] = qgten[k] - sed_g[k] * odzq * onstep[4] * orho;
        rg[k] = MAXD(R1, rg[k] - sed_g[k] * odzq * DT * onstep[4]);
        for (k = ksed1[4]; k >= kts; k--) {
            odzq = 1. / dzq[k];
            orho = 1. / rho[k];
            qgten[k] = qgten[k] + (sed_g[k + 1] - sed_g[k]) * odzq * onstep[4] * orho;
            rg[k] = MAXD(R1, rg[k] + (sed_g[k + 1] - sed_g[k]) * odzq * DT * onstep[4]);
        }
        if (rg[kts] > R1 * 10.)
            ppt[2] = ppt[2] + sed_g[kts] * DT * onstep[4];
    }

    /* ---- Q: instant melt / homogeneous freeze, M:3584-3606 ---- */
    if (!iiwarm) {
        for (k = kts; k <= kte; k++) {
            xri = MAXD(0.0, qi1d[k] + qiten[k] * DT);
            /* conditioning diagnostics (oracle only): the two `> 0.0` tests of this block are
             * taken on cancellation residues when the species was removed completely.  `force`
             * (oracle only, 0 = the reference's own decision) takes such a test as true (1) or
             * false (2): the two outcomes an implementation with other rounding may legitimately
             * produce at that level; block Q and block R are pointwise in k, so nothing else moves. */
            int res_ri = 0, res_rc = 0;
            {
                double sc = MAXD(fabs(qi1d[k]), fabs(qiten[k] * DT));
                if (temp[k] > T_0 && sc > 0. && fabs(qi1d[k] + qiten[k] * DT) <= 1e-9 * sc) res_ri = 1;
                sc = MAXD(fabs(qc1d[k]), fabs(qcten[k] * DT));
                /* xrc is re-evaluated after the melt branch; T > T_0 and T < HGFR exclude each other, so qcten
                 * is still the value this test will see */
                if (temp[k] < HGFR && sc > 0. && fabs(qc1d[k] + qcten[k] * DT) <= 1e-9 * sc) res_rc = 1;
                if (illcond) illcond[k] |= res_ri | (res_rc << 1);
            }
            const int take_ri = (force && res_ri) ? (force == 1) : (xri > 0.0);
            if ((temp[k] > T_0) && take_ri) {
                qcten[k] = qcten[k] + xri * odt;
                ncten[k] = ncten[k] + ni1d[k] * odt;
                qiten[k] = qiten[k] - xri * odt;
                niten[k] = -ni1d[k] * odt;
                tten[k] = tten[k] - lfus * ocp[k] * xri * odt * (1 - IFDRY);
            }

            xrc = MAXD(0.0, qc1d[k] + qcten[k] * DT);
            const int take_rc = (force && res_rc) ? (force == 1) : (xrc > 0.0);
            if ((temp[k] < HGFR) && take_rc) {
                lfus2 = lsub - lvap[k];
                xnc = nc1d[k] + ncten[k] * DT;
                qiten[k] = qiten[k] + xrc * odt;
                niten[k] = niten[k] + xnc * odt;
                qcten[k] = qcten[k] - xrc * odt;
                ncten[k] = ncten[k] - xnc * odt;
                tten[k] = tten[k] + lfus2 * ocp[k] * xrc * odt * (1 - IFDRY);
            }
        }
    }

    /* ---- R: apply tendencies, M:3623-3686 ---- */
    for (k = kts; k <= kte; k++) {
        t1d[k] = t1d[k] + tten[k] * DT;
        qv1d[k] = MAXD(1.E-10, qv1d[k] + qvten[k] * DT);
        qc1d[k] = qc1d[k] + qcten[k] * DT;
        nc1d[k] = MAXD(2. / rho[k], nc1d[k] + ncten[k] * DT);
        nwfa1d[k] = MAXD(11.1E6 / rho[k], MIND(9999.E6 / rho[k], (nwfa1d[k] + nwfaten[k] * DT)));
        nifa1d[k] = MAXD(naIN1 * 0.01, MIND(9999.E6 / rho[k], (nifa1d[k] + nifaten[k] * DT)));

        if (qc1d[k] <= R1) {
            qc1d[k] = 0.0;
            nc1d[k] = 0.0;
        } else {
            nu_c = NINT(1000.E6 / (nc1d[k] * rho[k])) + 2; if (nu_c > 15) nu_c = 15;
            lamc = pow(am_r * ccg2[nu_c] * ocg1[nu_c] * nc1d[k] / qc1d[k], obmr);
            xDc = (bm_r + nu_c + 1.) / lamc;
            if (xDc < D0c)
                lamc = cce2[nu_c] / D0c;
            else if (xDc > D0r * 2.)
                lamc = cce2[nu_c] / (D0r * 2.);
            nc1d[k] = MIND(ccg1[nu_c] * ocg2[nu_c] * qc1d[k] / am_r * pow(lamc, bm_r), Nt_c_max / rho[k]);
        }

        qi1d[k] = qi1d[k] + qiten[k] * DT;
        ni1d[k] = MAXD(R2 / rho[k], ni1d[k] + niten[k] * DT);
        if (qi1d[k] <= R1) {
            qi1d[k] = 0.0;
            ni1d[k] = 0.0;
        } else {
            lami = pow(am_i * cig[2] * oig1 * ni1d[k] / qi1d[k], obmi);
            ilami = 1. / lami;
            xDi = (bm_i + mu_i + 1.) * ilami;
            if (xDi < 5.E-6)
                lami = cie[2] / 5.E-6;
            else if (xDi > 300.E-6)
                lami = cie[2] / 300.E-6;
            ni1d[k] = MIND(cig[1] * oig2 * qi1d[k] / am_i * pow(lami, bm_i), (double)499.e3 / rho[k]);   /* 499.D3/rho(k), M:3664 */
        }
        qr1d[k] = qr1d[k] + qrten[k] * DT;
        nr1d[k] = MAXD(R2 / rho[k], nr1d[k] + nrten[k] * DT);
        if (qr1d[k] <= R1) {
            qr1d[k] = 0.0;
            nr1d[k] = 0.0;
        } else {
            lamr = pow(am_r * crg[3] * org2 * nr1d[k] / qr1d[k], obmr);
            mvd_r[k] = (3.0 + mu_r + 0.672) / lamr;
            if (mvd_r[k] > 2.5E-3)
                mvd_r[k] = 2.5E-3;
            else if (mvd_r[k] < D0r * 0.75)
                mvd_r[k] = D0r * 0.75;
            lamr = (3.0 + mu_r + 0.672) / mvd_r[k];
            nr1d[k] = crg[2] * org3 * qr1d[k] * pow(lamr, bm_r) / am_r;
        }
        qs1d[k] = qs1d[k] + qsten[k] * DT;
        if (qs1d[k] <= R1) qs1d[k] = 0.0;
        qg1d[k] = qg1d[k] + qgten[k] * DT;
        if (qg1d[k] <= R1) qg1d[k] = 0.0;
    }

    free(ws);
    free(wd);
    free(Lws);
    return 0;
}

/* calc_effectRad, M:4834-4935: radiation effective radii of cloud water, cloud ice and snow (INOUT: levels without
 * the species keep what the caller put there -- the scheme's driver presets 2.49E-6, 4.99E-6, 9.99E-6, M:1111-1113).
 * is_aerosol_aware = .false.: nc(k) = Nt_c (M:4863). */
void P(th_oracle_calc_effectRad)(const th_oracle *ctx, int nz,
                                 const real *t1d, const real *p1d, const real *qv1d, const real *qc1d,
                                 const real *nc1d, const real *qi1d, const real *ni1d, const real *qs1d,
                                 real *re_qc1d, real *re_qi1d, real *re_qs1d)
{
    const th_view *o = (const th_view *)ctx->P(view);
    static const real g_ratio[16] = { 0, 24, 60, 120, 210, 336, 504, 720, 990, 1320, 1716, 2184, 2730, 3360, 4080, 4896 };
    const real *cse = o->cse;
    for (int k = 0; k < nz; k++) {
        real rho = 0.622 * p1d[k] / (R_gas * t1d[k] * (qv1d[k] + 0.622));
        real rc = MAXD(R1, qc1d[k] * rho);
        real nc = MAXD(R2, nc1d[k] * rho);
        if (!o->is_aerosol_aware) nc = o->Nt_c;                         /* M:4863 */
        real ri = MAXD(R1, qi1d[k] * rho);
        real ni = MAXD(R2, ni1d[k] * rho);
        real rs = MAXD(R1, qs1d[k] * rho);
        if (!(rc <= R1 || nc <= R2)) {                                   /* M:4873-4884 */
            int inu_c;
            if (nc < 100) inu_c = 15;
            else if (nc > 1.E10) inu_c = 2;
            else { inu_c = NINT(1000.E6 / nc) + 2; if (inu_c > 15) inu_c = 15; }
            double lamc = pow(nc * am_r * g_ratio[inu_c] / rc, o->obmr);
            re_qc1d[k] = MAXD(2.51E-6, MIND((real)((double)0.5 * (double)(3. + inu_c) / lamc), 50.E-6));
        }
        if (!(ri <= R1 || ni <= R2)) {                                   /* M:4887-4893 */
            double lami = pow(am_i * o->cig[2] * o->oig1 * ni / ri, o->obmi);
            re_qi1d[k] = MAXD(5.01E-6, MIND((real)((double)0.5 * (double)(3. + mu_i) / lami), 125.E-6));
        }
        if (!(rs <= R1)) {                                               /* M:4896-4930; bm_s = 2: smo2 = smob */
            real tc0 = MIND(-0.1, t1d[k] - 273.15);
            real smob = rs * o->oams;
            real smo2 = smob;
            real loga_ = mom_loga(o->sa, tc0, cse[1]);
            real a_ = pow((real)10.0, loga_);
            real b_ = mom_b(o->sb, tc0, cse[1]);
            real smoc = a_ * pow(smo2, b_);
            re_qs1d[k] = MAXD(10.E-6, MIND(0.5 * (smoc / smob), 999.E-6));
        }
    }
}

/* non-aerosol defaults, M:958-964 (decision U2) */
void P(th_oracle_default_aerosols)(const th_oracle *ctx, int nz,
                                const real *qv1d, const real *t1d,
                                const real *p1d, real *nc1d,
                                real *nwfa1d, real *nifa1d)
{
    const th_view *o = (const th_view *)ctx->P(view);
    for (int k = 0; k < nz; k++) {
        real rho = 0.622 * p1d[k] / (R_gas * t1d[k] * (qv1d[k] + 0.622));
        nc1d[k] = o->Nt_c / rho;
        nwfa1d[k] = 11.1E6 / rho;
        nifa1d[k] = naIN1 * 0.01 / rho;
    }
}

/* ------------------------------------------------------------------ */
typedef struct {
    const th_oracle *o; long ncol, per; int nz; real dt;
    real *qv, *qc, *qi, *qr, *qs, *qg, *ni, *nr, *nc, *nwfa, *nifa, *t;
    const real *p, *w, *dz; real *ppt; int *illcond; int force;
} batch_job;

/* one chunk = `per` consecutive columns (handed out dynamically by the pool) */
static void batch_chunk(void *arg, long chunk)
{
    batch_job *b = (batch_job *)arg;
    const size_t nz = (size_t)b->nz;
    const long c0 = chunk * b->per, c1 = c0 + b->per < b->ncol ? c0 + b->per : b->ncol;
    for (long c = c0; c < c1; c++) {
        size_t off = (size_t)c * nz;
        P(th_oracle_mp_thompson_force)(b->o, b->qv + off, b->qc + off, b->qi + off, b->qr + off,
                                 b->qs + off, b->qg + off, b->ni + off, b->nr + off,
                                 b->nc + off, b->nwfa + off, b->nifa + off, b->t + off,
                                 b->p + off, b->w + off, b->dz + off, b->ppt + 4 * (size_t)c,
                                 b->nz, b->dt, NULL, NULL, b->illcond ? b->illcond + off : NULL, b->force);
    }
}

#ifndef TH_P32N
int th_oracle_batch(const th_oracle *o, long ncol, int nz, double dt,
                    double *qv, double *qc, double *qi, double *qr,
                    double *qs, double *qg, double *ni, double *nr,
                    double *nc, double *nwfa, double *nifa, double *t,
                    const double *p, const double *w, const double *dz,
                    double *ppt, int nthreads)
{
    return th_oracle_batch_ex(o, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt,
                              nthreads, NULL);
}

int th_oracle_batch_ex(const th_oracle *o, long ncol, int nz, double dt,
                       double *qv, double *qc, double *qi, double *qr,
                       double *qs, double *qg, double *ni, double *nr,
                       double *nc, double *nwfa, double *nifa, double *t,
                       const double *p, const double *w, const double *dz,
                       double *ppt, int nthreads, int *illcond)
{
    return th_oracle_batch_force(o, ncol, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt,
                                 nthreads, illcond, 0);
}

#endif

int P(th_oracle_batch_force)(const th_oracle *o, long ncol, int nz, real dt,
                          real *qv, real *qc, real *qi, real *qr,
                          real *qs, real *qg, real *ni, real *nr,
                          real *nc, real *nwfa, real *nifa, real *t,
                          const real *p, const real *w, const real *dz,
                          real *ppt, int nthreads, int *illcond, int force)
{
    if (nthreads < 1) nthreads = 1;
    if (ncol <= 0) return 0;
    /* columns go to the persistent pool in chunks of `per`, handed out dynamically: about 8 chunks per thread, so that
     * columns of unequal cost (substep counts, regimes) balance, and at least 4 columns per chunk */
    long per = ncol / ((long)nthreads * 8);
    if (per < 4) per = 4;
    if (per > 64) per = 64;
    batch_job b = { o, ncol, per, nz, dt, qv, qc, qi, qr, qs, qg, ni, nr, nc, nwfa, nifa, t, p, w, dz, ppt, illcond, force };
    th_pool_run(nthreads, batch_chunk, &b, (ncol + per - 1) / per);
    return 0;
}

/* ------------------------------------------------------------------ */
/* mphys_thompson09_interfacen, W:28-310 */
int P(th_oracle_kid_interface)(const th_oracle *o, int nz, int nx, real dt,
                            real p0, real r_on_cp,
                            const real *theta, const real *dtheta_adv,
                            const real *dtheta_div, const real *exner,
                            const real *dz, const real *qv,
                            const real *dqv_adv, const real *dqv_div,
                            const real *hydro, const real *dhydro_adv,
                            const real *dhydro_div,
                            real *dtheta_mphys, real *dqv_mphys,
                            real *dhydro_mphys, real *ppt)
{
    const size_t N = (size_t)nz;
    real *buf = (real *)calloc(16 * N, sizeof(real));
    if (!buf) return -1;
    real *t1d = buf, *p1d = buf + N, *dzq = buf + 2 * N, *qv1d = buf + 3 * N, *qc1d = buf + 4 * N,
           *qr1d = buf + 5 * N, *qi1d = buf + 6 * N, *ni1d = buf + 7 * N, *qs1d = buf + 8 * N,
           *qg1d = buf + 9 * N, *nr1d = buf + 10 * N, *nc1d = buf + 11 * N, *nifa1d = buf + 12 * N,
           *nwfa1d = buf + 13 * N, *w1d = buf + 14 * N;
#define H(a, k, i, ih, im) (a)[(size_t)(k) + N * ((size_t)(i) + (size_t)nx * ((size_t)(ih) + 5 * (size_t)(im)))]
#define S(a, k, i)         (a)[(size_t)(k) + N * (size_t)(i)]
    /* qc1d..qg1d zero-initialised once, outside the i loop (W:46-52) */
    for (int i = 0; i < nx; i++) {
        real pp[4] = { 0., 0., 0., 0. };                    /* W:55-58 */
        for (int k = 0; k < nz; k++) {                         /* W:59-97 */
            t1d[k] = (S(theta, k, i) + (S(dtheta_adv, k, i) + S(dtheta_div, k, i)) * dt) * S(exner, k, i);
            p1d[k] = p0 * pow(S(exner, k, i), 1. / r_on_cp);
            dzq[k] = dz[k];
            qv1d[k] = S(qv, k, i) + (S(dqv_adv, k, i) + S(dqv_div, k, i)) * dt;
            qc1d[k] = H(hydro, k, i, 0, 0) + (H(dhydro_adv, k, i, 0, 0) + H(dhydro_div, k, i, 0, 0)) * dt;
            qr1d[k] = H(hydro, k, i, 1, 0) + (H(dhydro_adv, k, i, 1, 0) + H(dhydro_div, k, i, 1, 0)) * dt;
            nr1d[k] = H(hydro, k, i, 1, 1) + (H(dhydro_adv, k, i, 1, 1) + H(dhydro_div, k, i, 1, 1)) * dt;
            if (!o->iiwarm) {
                qi1d[k] = H(hydro, k, i, 2, 0) + (H(dhydro_adv, k, i, 2, 0) + H(dhydro_div, k, i, 2, 0)) * dt;
                ni1d[k] = H(hydro, k, i, 2, 1) + (H(dhydro_adv, k, i, 2, 1) + H(dhydro_div, k, i, 2, 1)) * dt;
                qs1d[k] = H(hydro, k, i, 3, 0) + (H(dhydro_adv, k, i, 3, 0) + H(dhydro_div, k, i, 3, 0)) * dt;
                qg1d[k] = H(hydro, k, i, 4, 0) + (H(dhydro_adv, k, i, 4, 0) + H(dhydro_div, k, i, 4, 0)) * dt;
            }
            w1d[k] = 0.;
        }
        /* U2: the wrapper leaves nc1d, nwfa1d, nifa1d, w1d unset (W:36) */
        P(th_oracle_default_aerosols)(o, nz, qv1d, t1d, p1d, nc1d, nwfa1d, nifa1d);

        P(th_oracle_mp_thompson_force)(o, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d, nifa1d,
                              t1d, p1d, w1d, dzq, pp, nz, dt, NULL, NULL, NULL, 0);     /* W:143-152 */

        ppt[0 * (size_t)nx + i] = pp[0];
        ppt[1 * (size_t)nx + i] = pp[1];
        ppt[2 * (size_t)nx + i] = pp[2];
        ppt[3 * (size_t)nx + i] = pp[3];

        for (int k = 0; k < nz; k++) {                         /* W:198-245 */
            S(dtheta_mphys, k, i) = (t1d[k] / S(exner, k, i) - S(theta, k, i)) / dt
                                  - (S(dtheta_adv, k, i) + S(dtheta_div, k, i));
            S(dqv_mphys, k, i) = (qv1d[k] - S(qv, k, i)) / dt - (S(dqv_adv, k, i) + S(dqv_div, k, i));
            H(dhydro_mphys, k, i, 0, 0) = (qc1d[k] - H(hydro, k, i, 0, 0)) / dt
                                        - (H(dhydro_adv, k, i, 0, 0) + H(dhydro_div, k, i, 0, 0));
            H(dhydro_mphys, k, i, 1, 0) = (qr1d[k] - H(hydro, k, i, 1, 0)) / dt
                                        - (H(dhydro_adv, k, i, 1, 0) + H(dhydro_div, k, i, 1, 0));
            H(dhydro_mphys, k, i, 1, 1) = (nr1d[k] - H(hydro, k, i, 1, 1)) / dt
                                        - (H(dhydro_adv, k, i, 1, 1) + H(dhydro_div, k, i, 1, 1));
            if (!o->iiwarm) {
                H(dhydro_mphys, k, i, 2, 0) = (qi1d[k] - H(hydro, k, i, 2, 0)) / dt
                                            - (H(dhydro_adv, k, i, 2, 0) + H(dhydro_div, k, i, 2, 0));
                H(dhydro_mphys, k, i, 2, 1) = (ni1d[k] - H(hydro, k, i, 2, 1)) / dt
                                            - (H(dhydro_adv, k, i, 2, 1) + H(dhydro_div, k, i, 2, 1));
                H(dhydro_mphys, k, i, 3, 0) = (qs1d[k] - H(hydro, k, i, 3, 0)) / dt
                                            - (H(dhydro_adv, k, i, 3, 0) + H(dhydro_div, k, i, 3, 0));
                H(dhydro_mphys, k, i, 4, 0) = (qg1d[k] - H(hydro, k, i, 4, 0)) / dt
                                            - (H(dhydro_adv, k, i, 4, 0) + H(dhydro_div, k, i, 4, 0));
            }
        }
    }
#undef H
#undef S
    free(buf);
    return 0;
}
