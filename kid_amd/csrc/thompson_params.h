// thompson_params.h -- constants and table handles shared by the host-side
// initialiser and the gfx950 kernels.  All values are what the reference's
// module header and thompson_init compute (M:25-363, M:374-602), P64 arithmetic.
#pragma once
#include <cstdint>

namespace kidmp {

// ---- compile-time PARAMETERs of module_mp_thompson09n (M:30-204) ----
constexpr double T_0 = 273.15;
constexpr double PI = 3.1415926536;            // sic, 10 digits (M:35)
constexpr double rho_w = 1000.0, rho_s = 100.0, rho_g = 500.0, rho_i = 890.0;
constexpr double Nt_c_max = 1999.E6;
constexpr double naIN1 = 0.5E6;
constexpr double mu_r = 0.0, mu_g = 0.0, mu_i = 0.0, mu_s = 0.6357;
constexpr double Kap0 = 490.6, Kap1 = 17.46, Lam0 = 20.78, Lam1 = 3.29;
constexpr double gonv_min = 1.E4, gonv_max = 3.E6;
constexpr double am_r = PI * rho_w / 6.0, bm_r = 3.0;
constexpr double am_s = 0.069, bm_s = 2.0;
constexpr double am_g = PI * rho_g / 6.0, bm_g = 3.0;
constexpr double am_i = PI * rho_i / 6.0, bm_i = 3.0;
constexpr double av_r = 4854.0, bv_r = 1.0, fv_r = 195.0;
constexpr double av_s = 40.0, bv_s = 0.55, fv_s = 100.0;
constexpr double av_g = 442.0, bv_g = 0.89;
constexpr double av_i = 1847.5, bv_i = 1.0;
constexpr double bv_c = 2.0;
constexpr double C_cube = 0.5, C_sqrd = 0.15;
constexpr double Ef_si = 0.05, Ef_rs = 0.95, Ef_rg = 0.75, Ef_ri = 0.95;
constexpr double R1 = 1.E-12, R2 = 1.E-6, eps = 1.E-15;
constexpr double TNO = 5.0, ATO = 0.304;
constexpr double rho_not = 101325.0 / (287.05 * 298.0);
constexpr double Sc = 0.632;
constexpr double HGFR = 235.16;
constexpr double Rv = 461.5, oRv = 1. / Rv, Rgas = 287.04, Cp = 1004.0;
constexpr double lsub = 2.834E6, lvap0 = 2.5E6, lfus = lsub - lvap0, olfus = 1. / lfus;
constexpr double xm0i = 1.E-12, D0c = 1.E-6, D0r = 50.E-6, D0s = 200.E-6, D0g = 250.E-6;

constexpr int nbins = 100;
constexpr int ntb_c = 37, ntb_i = 64, ntb_r = 37, ntb_s = 28, ntb_g = 28, ntb_g1 = 28,
              ntb_r1 = 37, ntb_i1 = 55, ntb_t = 9, ntb_IN = 55, ntb_tc = 45;

// table sizes (doubles)
constexpr int64_t N_RACG = int64_t(ntb_g1) * ntb_g * ntb_r1 * ntb_r;
constexpr int64_t N_RACS = int64_t(ntb_s) * ntb_t * ntb_r1 * ntb_r;
constexpr int64_t N_QRFZ = int64_t(ntb_r) * ntb_r1 * ntb_tc;
constexpr int64_t N_QCFZ = int64_t(ntb_c) * ntb_tc;
constexpr int64_t N_IAUS = int64_t(ntb_i) * ntb_i1;
constexpr int64_t N_EF = int64_t(nbins) * nbins;
constexpr int64_t N_WEV = int64_t(nbins) * ntb_c * nbins;

// ---- values thompson_init computes once (M:442-602); 0-based C arrays hold
// the Fortran element n at index n-1 ----
struct Consts {
    int32_t iiwarm, l_sediment;
    double Nt_c;
    double Sc3, D0i, xm0s, xm0g;
    double cce[5][15], ccg[5][15], ocg1[15], ocg2[15];
    double cie[7], cig[7], oig1, oig2, obmi;
    double cre[13], crg[13], ore1, org1, org2, org3, obmr;
    double cse[18], csg[18], oams, obms, ocms;
    double cge[12], cgg[12], oge1, ogg1, ogg2, ogg3, oamg, obmg, ocmg;
    double t1_qr_qc, t1_qr_qi, t2_qr_qi, t1_qg_qc, t1_qs_qc, t1_qs_qi;
    double t1_qr_ev, t2_qr_ev, t1_qs_sd, t2_qs_sd, t1_qg_sd, t2_qg_sd;
    double t1_qs_me, t2_qs_me, t1_qg_me, t2_qg_me;
    int32_t nic2, nii2, nii3, nir2, nir3, nis2, nig2, nig3;
    // constant sub-expressions the solver evaluates at every level (hoisted to init):
    double lamg_fac;       // (cgg(3)*ogg2*ogg1)**obmg          M:1651, M:2735
    double lamr_exp_fac;   // (crg(3)*org2*org1)**bm_r          M:1823
    double lamg_exp_fac;   // (cgg(3)*ogg2*ogg1)**bm_g          M:1867
    double dcg_fac[15];    // (ccg(3,nu_c)*ocg2(nu_c))**obmr    M:1699
    double sa[10], sb[10];
    // first/last bin centres used by the efficiency-table index (M:1717, M:1907)
    double Dr1, Drn, Ds1, Dsn;
    // axis minima used as thresholds in the solver
    double r_c1, r_i1, r_r1, r_s1, r_g1, Nt_i1;
    // droplet-number axis of tnc_wev (aerosol-aware droplet evaporation, M:2828): t_Nc(1) and the INTEGER nic1 of M:670
    double t_Nc1;
    int32_t nic1, pad_;
    // what the column kernel copies into LDS at workgroup start, as one flat table (one load per thread):
    // [0..111]  seven rows of 16 indexed by nu_c-1 (entry 15 repeats 14): ccg(1,:), ccg(2,:), ocg1, ocg2, cce(2,:), dcg_fac, (ccg(2,:)*ocg1)**obmr
    // [112..143] 10**-n for n = -16 .. 15: the decade finder's scale factors (exact powers of ten, or their
    //           correctly rounded reciprocals)
    double lds_tab[144];
    // graupel intercept (M:1639-1647) of a level without graupel (rg <= 5.E-5) and without supercooled rain above k_0:
    // a constant of the scheme, evaluated once per arithmetic variant ON THE DEVICE by the kernel's own function
    // (upload_consts), so that it carries exactly the bits the per-level evaluation would produce
    double n0g_empty;
};

// bins and axes needed only while building tables (device copies)
struct Bins {
    double Dc[nbins], dtc[nbins], Di[nbins], dti[nbins], Dr[nbins], dtr[nbins],
           Ds[nbins], dts[nbins], Dg[nbins], dtg[nbins], t_Nc[nbins];
    double r_c[ntb_c], r_i[ntb_i], r_r[ntb_r], r_g[ntb_g], r_s[ntb_s],
           N0r_exp[ntb_r1], N0g_exp[ntb_g1], Nt_i[ntb_i1], Nt_IN[ntb_IN], Tc[ntb_t];
};

// ---- lookup tables resident in HBM (Fortran column-major order kept).
// The rain-snow and rain-graupel families are additionally stored interleaved
// per cell (one record per (i,j,k,m)) so that the <=10 / <=5 values a level
// needs come from one or two cache lines instead of 10 / 5 separate gathers.
struct Tables {
    // planar copies (parity tests read these; M:324-338 names)
    double *tcg_racg, *tmr_racg, *tcr_gacr, *tmg_gacr, *tnr_racg, *tnr_gacr;
    double *tcs_racs1, *tmr_racs1, *tcs_racs2, *tmr_racs2, *tcr_sacr1, *tms_sacr1,
           *tcr_sacr2, *tms_sacr2, *tnr_racs1, *tnr_racs2, *tnr_sacr1, *tnr_sacr2;
    double *tpi_qcfz, *tni_qcfz;
    double *tpi_qrfz, *tpg_qrfz, *tni_qrfz, *tnr_qrfz;
    double *tps_iaus, *tni_iaus, *tpi_ide;
    double *t_Efrw, *t_Efsw;
    double *tnc_wev;    // table_dropEvap M:4400-4439, (nbc, ntb_c, nbc): read only by aerosol-aware contexts (M:2850)
    // interleaved records read by the column kernel
    double *racs_rec;   // [N_RACS][10]: tmr_racs1,tcr_sacr1,tmr_racs2,tcr_sacr2,tcs_racs1,tms_sacr1,tnr_racs1,tnr_racs2,tnr_sacr1,tnr_sacr2
    double *racg_rec;   // [N_RACG][5] : tmr_racg,tcr_gacr,tnr_racg,tnr_gacr,tcg_racg
    double *qrfz_rec;   // [N_QRFZ][4] : tpg,tpi,tni,tnr
};
constexpr int RACS_REC = 10, RACG_REC = 5, QRFZ_REC = 4;

}  // namespace kidmp
