/*
 * thompson_oracle_internal.h -- CPU ORACLE internals (test infrastructure).
 * Module-level state of module_mp_thompson09n (M:25-363) as one struct.
 * All arrays that the Fortran indexes from 1 are declared one element longer
 * and indexed from 1 here, so the restated formulas read like the source.
 */
#ifndef THOMPSON_ORACLE_INTERNAL_H
#define THOMPSON_ORACLE_INTERNAL_H

#include "thompson_oracle.h"
#include <math.h>
#include <stddef.h>

/* ---- PARAMETERs, M:30-204 (P64: literals are binary64) ---- */
#define T_0      273.15
#define PI       3.1415926536
#define rho_w    1000.0
#define rho_s    100.0
#define rho_g    500.0
#define rho_i    890.0
#define Nt_c_max 1999.E6
#define naIN0    1.5E6
#define naIN1    0.5E6
#define mu_r     0.0
#define mu_g     0.0
#define mu_i     0.0
#define mu_s     0.6357
#define Kap0     490.6
#define Kap1     17.46
#define Lam0     20.78
#define Lam1     3.29
#define gonv_min 1.E4
#define gonv_max 3.E6
#define am_r     (PI*rho_w/6.0)
#define bm_r     3.0
#define am_s     0.069
#define bm_s     2.0
#define am_g     (PI*rho_g/6.0)
#define bm_g     3.0
#define am_i     (PI*rho_i/6.0)
#define bm_i     3.0
#define av_r     4854.0
#define bv_r     1.0
#define fv_r     195.0
#define av_s     40.0
#define bv_s     0.55
#define fv_s     100.0
#define av_g     442.0
#define bv_g     0.89
#define av_i     1847.5
#define bv_i     1.0
#define av_c     0.316946E8
#define bv_c     2.0
#define C_cube   0.5
#define C_sqrd   0.15
#define Ef_si    0.05
#define Ef_rs    0.95
#define Ef_rg    0.75
#define Ef_ri    0.95
#define R1       1.E-12
#define R2       1.E-6
#define eps      1.E-15
#define TNO      5.0
#define ATO      0.304
#define rho_not  (101325.0/(287.05*298.0))
#define Sc       0.632
#define HGFR     235.16
#define Rv       461.5
#define oRv      (1./Rv)
#define R_gas    287.04          /* "R" at M:153 */
#define Cp       1004.0
#define lsub     2.834E6
#define lvap0    2.5E6
#define lfus     (lsub - lvap0)
#define olfus    (1./lfus)
#define xm0i     1.E-12
#define D0c      1.E-6
#define D0r      50.E-6
#define D0s      200.E-6
#define D0g      250.E-6
#define IFDRY    0

enum { nbins = 100, nbc = 100, nbi = 100, nbr = 100, nbs = 100, nbg = 100,
       ntb_c = 37, ntb_i = 64, ntb_r = 37, ntb_s = 28, ntb_g = 28,
       ntb_g1 = 28, ntb_r1 = 37, ntb_i1 = 55, ntb_t = 9, ntb_IN = 55 };

#define MAXD(a,b) ((a) > (b) ? (a) : (b))
#define MIND(a,b) ((a) < (b) ? (a) : (b))
#define NINT(x)   ((int)lround(x))      /* Fortran NINT: half away from 0 */

struct th_oracle {
    /* switches / namelist values */
    int iiwarm, l_sediment;
    double set_Nc, Nt_c;
    /* M:145,177 */
    double Sc3, D0i, xm0s, xm0g;
    /* axis vectors M:215-315 (1-based) */
    double r_c[ntb_c+1], r_i[ntb_i+1], r_r[ntb_r+1], r_g[ntb_g+1],
           r_s[ntb_s+1], N0r_exp[ntb_r1+1], N0g_exp[ntb_g1+1],
           Nt_i[ntb_i1+1], Nt_IN[ntb_IN+1], sa[11], sb[11], Tc[ntb_t+1];
    /* gamma-function constants M:346-355 (1-based) */
    double cce[6][16], ccg[6][16], ocg1[16], ocg2[16];
    double cie[8], cig[8], oig1, oig2, obmi;
    double cre[14], crg[14], ore1, org1, org2, org3, obmr;
    double cse[19], csg[19], oams, obms, ocms;
    double cge[13], cgg[13], oge1, ogg1, ogg2, ogg3, oamg, obmg, ocmg;
    /* rate prefactors M:358-361 */
    double t1_qr_qc, t1_qr_qi, t2_qr_qi, t1_qg_qc, t1_qs_qc, t1_qs_qi;
    double t1_qr_ev, t2_qr_ev;
    double t1_qs_sd, t2_qs_sd, t1_qg_sd, t2_qg_sd;
    double t1_qs_me, t2_qs_me, t1_qg_me, t2_qg_me;
    /* index offsets M:195,202 */
    int nic1, nic2, nii2, nii3, nir2, nir3, nis2, nig2, nig3, niIN2;
    /* bins M:206-212 (1-based) */
    double Dc[nbc+1], dtc[nbc+1], Di[nbi+1], dti[nbi+1], Dr[nbr+1],
           dtr[nbr+1], Ds[nbs+1], dts[nbs+1], Dg[nbg+1], dtg[nbg+1],
           t_Nc[nbc+1];
    /* lookup tables M:324-338, Fortran (column-major) order, 0-based flat */
    double *tcg_racg, *tmr_racg, *tcr_gacr, *tmg_gacr, *tnr_racg, *tnr_gacr;
    double *tcs_racs1, *tmr_racs1, *tcs_racs2, *tmr_racs2, *tcr_sacr1,
           *tms_sacr1, *tcr_sacr2, *tms_sacr2, *tnr_racs1, *tnr_racs2,
           *tnr_sacr1, *tnr_sacr2;
    double *tpi_qcfz, *tni_qcfz;
    double *tpi_qrfz, *tpg_qrfz, *tni_qrfz, *tnr_qrfz;
    double *tps_iaus, *tni_iaus, *tpi_ide;
    double *t_Efrw, *t_Efsw;
    double *tnc_wev, *tpc_wev;        /* table_dropEvap M:4400-4439, (nbc, ntb_c, nbc): only read when is_aerosol_aware */
    int is_aerosol_aware;             /* M:28 (.false. in KiD); th_oracle_set_aerosol_aware */
    int nthreads;
    /* what mp_thompson reads, in the arithmetic of each build of thompson_oracle_column.c (th_view there) */
    void *view, *view_p32n;
};

void *th_oracle_make_view(const struct th_oracle *c);        /* P64 build of the column file */
void *th_oracle_make_view_p32n(const struct th_oracle *c);   /* P32n build (thompson_oracle_p32n.c) */

/* Persistent worker pool shared by both arithmetic builds (defined once, thompson_oracle_init.c).
 * th_pool_run calls fn(arg, chunk) for every chunk in [0, nchunks) on `nthreads` threads (the caller is one of
 * them); chunks are handed out dynamically.  Threads are created on first use and kept, so a timed region made of
 * many batch calls pays no pthread_create/join per call.  Calls are serialised by a mutex. */
void th_pool_run(int nthreads, void (*fn)(void *arg, long chunk), void *arg, long nchunks);

/* column-major index helpers (1-based arguments) */
#define IX2(i,j,n1)             ((size_t)((i)-1) + (size_t)(n1)*((j)-1))
#define IX3(i,j,k,n1,n2)        ((size_t)((i)-1) + (size_t)(n1)*(((j)-1) + (size_t)(n2)*((k)-1)))
#define IX4(i,j,k,m,n1,n2,n3)   ((size_t)((i)-1) + (size_t)(n1)*(((j)-1) + (size_t)(n2)*(((k)-1) + (size_t)(n3)*((m)-1))))
#define RACG(t,i,j,k,m)  (t)[IX4(i,j,k,m,ntb_g1,ntb_g,ntb_r1)]
#define RACS(t,i,j,k,m)  (t)[IX4(i,j,k,m,ntb_s,ntb_t,ntb_r1)]
#define QRFZ(t,i,j,k)    (t)[IX3(i,j,k,ntb_r,ntb_r1)]
#define QCFZ(t,i,k)      (t)[IX2(i,k,ntb_c)]
#define IAUS(t,i,j)      (t)[IX2(i,j,ntb_i)]
#define EFRW(t,i,j)      (t)[IX2(i,j,nbr)]
#define EFSW(t,i,j)      (t)[IX2(i,j,nbs)]

/* real**integer as flang lowers it (compiler-rt __powidf2), e.g. 10.**nn at
 * M:1766: square-and-multiply, reciprocal at the end for negative n. */
static inline double th_powi(double a, int b)
{
    const int recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

#endif
