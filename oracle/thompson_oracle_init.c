/*
 * thompson_oracle_init.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restates thompson_init (M:374-797), the table builders (M:3698-4343) and the
 * Numerical-Recipes gamma helpers (M:4530-4651) of
 * /root/reference/module_mp_thompson09n.f90 in plain C, P64 arithmetic.
 * Dead-with-aerosol-off pieces (table_dropEvap M:4400-4439, tnr_rev,
 * tnccn_act) are not built: nothing reads them when is_aerosol_aware=.false.
 */
#include "thompson_oracle_internal.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------ */
/* Persistent worker pool (see thompson_oracle_internal.h).  Test infrastructure: it exists so that the CPU baseline
 * of bench.py measures the restated physics, not pthread_create. */
#define TH_POOL_MAX 1024
static struct {
    pthread_mutex_t run_mu;            /* one th_pool_run at a time */
    pthread_mutex_t mu;
    pthread_cond_t wake, done;
    pthread_t th[TH_POOL_MAX];
    int nworkers;                      /* threads created so far (the caller is an extra one) */
    unsigned long gen;                 /* job generation */
    int want;                          /* workers that take part in the current job */
    int running;                       /* workers still inside the current job */
    void (*fn)(void *, long);
    void *arg;
    long nchunks;
    long next;                         /* next chunk to hand out (atomic) */
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER,
             {0}, 0, 0, 0, 0, NULL, NULL, 0, 0 };

static void pool_drain(void)
{
    for (;;) {
        long c = __atomic_fetch_add(&g_pool.next, 1, __ATOMIC_RELAXED);
        if (c >= g_pool.nchunks) break;
        g_pool.fn(g_pool.arg, c);
    }
}

static void *pool_worker(void *idp)
{
    const int id = (int)(size_t)idp;
    unsigned long seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.wake, &g_pool.mu);
        seen = g_pool.gen;
        if (id >= g_pool.want) continue;            /* not part of this job */
        pthread_mutex_unlock(&g_pool.mu);
        pool_drain();
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.running == 0) pthread_cond_signal(&g_pool.done);
    }
    return NULL;
}

/* a forked child inherits the pool's bookkeeping but none of its threads: start over there */
static void pool_reset_in_child(void)
{
    pthread_mutex_init(&g_pool.run_mu, NULL);
    pthread_mutex_init(&g_pool.mu, NULL);
    pthread_cond_init(&g_pool.wake, NULL);
    pthread_cond_init(&g_pool.done, NULL);
    g_pool.nworkers = 0; g_pool.gen = 0; g_pool.want = 0; g_pool.running = 0;
}
static pthread_once_t g_pool_once = PTHREAD_ONCE_INIT;
static void pool_register_atfork(void) { pthread_atfork(NULL, NULL, pool_reset_in_child); }

void th_pool_run(int nthreads, void (*fn)(void *arg, long chunk), void *arg, long nchunks)
{
    if (nchunks <= 0) return;
    pthread_once(&g_pool_once, pool_register_atfork);
    if (nthreads > nchunks) nthreads = (int)nchunks;
    if (nthreads > TH_POOL_MAX + 1) nthreads = TH_POOL_MAX + 1;
    if (nthreads <= 1) {
        for (long c = 0; c < nchunks; c++) fn(arg, c);
        return;
    }
    pthread_mutex_lock(&g_pool.run_mu);
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.nworkers < nthreads - 1) {        /* grow the pool; workers are daemons for the life of the process */
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        if (pthread_create(&g_pool.th[g_pool.nworkers], &at, pool_worker, (void *)(size_t)g_pool.nworkers) != 0) {
            pthread_attr_destroy(&at);
            break;
        }
        pthread_attr_destroy(&at);
        g_pool.nworkers++;
    }
    const int helpers = g_pool.nworkers < nthreads - 1 ? g_pool.nworkers : nthreads - 1;
    g_pool.fn = fn; g_pool.arg = arg; g_pool.nchunks = nchunks;
    __atomic_store_n(&g_pool.next, 0, __ATOMIC_RELAXED);
    g_pool.want = helpers; g_pool.running = helpers;
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.wake);
    pthread_mutex_unlock(&g_pool.mu);
    pool_drain();                                   /* the caller works too */
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.running > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
    pthread_mutex_unlock(&g_pool.run_mu);
}


/* ------------------------------------------------------------------ */
/* GAMMLN M:4598-4620 (6-term Lanczos; returns default REAL = fp64 in P64) */
double th_oracle_gammln(double xx)
{
    static const double STP = 2.5066282746310005;
    static const double COF[6] = { 76.18009172947146, -86.50532032941677,
                                   24.01409824083091, -1.231739572450155,
                                   .1208650973866179e-2, -.5395239384953e-5 };
    double x = xx, y = x, tmp = x + 5.5, ser;
    tmp = (x + 0.5) * log(tmp) - tmp;
    ser = 1.000000000190015;
    for (int j = 0; j < 6; j++) { y = y + 1.0; ser = ser + COF[j] / y; }
    return tmp + log(STP * ser / x);
}
/* WGAMMA M:4644-4651 */
static double wgamma(double y) { return exp(th_oracle_gammln(y)); }

/* GSER M:4566-4595 */
static double gser(double a, double x)
{
    const int ITMAX = 100; const double gEPS = 3.E-7;
    double gln = th_oracle_gammln(a);
    if (x <= 0.) return 0.;
    double ap = a, sum = 1. / a, del = sum;
    for (int n = 1; n <= ITMAX; n++) {
        ap = ap + 1.;
        del = del * x / ap;
        sum = sum + del;
        if (fabs(del) < fabs(sum) * gEPS) break;
    }
    return sum * exp(-x + a * log(x) - gln);
}
/* GCF M:4530-4563 (modified Lentz) */
static double gcf(double a, double x)
{
    const int ITMAX = 100; const double gEPS = 3.E-7, FPMIN = 1.E-30;
    double gln = th_oracle_gammln(a);
    double b = x + 1. - a, c = 1. / FPMIN, d = 1. / b, h = d, an, del;
    for (int i = 1; i <= ITMAX; i++) {
        an = -i * (i - a);
        b = b + 2.;
        d = an * d + b;
        if (fabs(d) < FPMIN) d = FPMIN;
        c = b + an / c;
        if (fabs(c) < FPMIN) c = FPMIN;
        d = 1. / d;
        del = d * c;
        h = h * del;
        if (fabs(del - 1.) < gEPS) break;
    }
    return exp(-x + a * log(x) - gln) * h;
}
/* GAMMP M:4623-4641 */
double th_oracle_gammp(double a, double x)
{
    if (x < 0. || a <= 0.) return 0.;
    if (x < a + 1.) return gser(a, x);
    return 1. - gcf(a, x);
}

/* ------------------------------------------------------------------ */
static void fill_decades(double *v, int n, double first)
{
    /* "1..9 per decade" axes, M:215-303.  The literals are decimal text
     * (1.e-6, 2.e-6 ...), so every entry must be the binary64 nearest to that
     * text, not a product: format and re-parse. */
    char buf[32];
    int e = NINT(log10(first));
    int idx = 1;
    for (int d = 0; idx <= n; d++)
        for (int m = 1; m <= 9 && idx <= n; m++) {
            snprintf(buf, sizeof buf, "%d.e%d", m, e + d);
            v[idx++] = strtod(buf, NULL);
        }
}

static double *talloc(size_t n)
{
    double *p = (double *)calloc(n, sizeof(double));   /* zeroed, M:676-742 */
    if (!p) { fprintf(stderr, "thompson_oracle: out of memory\n"); abort(); }
    return p;
}

/* log-spaced bins, M:612-658: xDx(n)=DEXP(DFLOAT(n-1)/DFLOAT(nb)*DLOG(hi/lo)+DLOG(lo)) */
static void make_bins(double lo, double hi, int nb, double *D, double *dt)
{
    double xDx[nbins + 2];
    xDx[1] = lo;
    xDx[nb + 1] = hi;
    for (int n = 2; n <= nb; n++)
        xDx[n] = exp((double)(n - 1) / (double)nb * log(xDx[nb + 1] / xDx[1]) + log(xDx[1]));
    for (int n = 1; n <= nb; n++) {
        D[n] = sqrt(xDx[n] * xDx[n + 1]);
        if (dt) dt[n] = xDx[n + 1] - xDx[n];
    }
}

/* ------------------------------------------------------------------ */
/* table_Efrw M:4243-4299 */
static void table_Efrw(th_oracle *o)
{
    for (int j = 1; j <= nbc; j++)
        for (int i = 1; i <= nbr; i++) {
            double Ef_rw = 0.0, vtr, stokes, reynolds, yc0, F, G, H, z, K0, X;
            const double Dri = o->Dr[i], Dcj = o->Dc[j];
            double p = Dcj / Dri;
            if (Dri < 50.E-6 || Dcj < 3.E-6) {
                EFRW(o->t_Efrw, i, j) = 0.0;
            } else if (p > 0.25) {
                X = Dcj * 1.e6;
                if (Dri < 75.e-6)
                    Ef_rw = 0.026794 * X - 0.20604;
                else if (Dri < 125.e-6)
                    Ef_rw = -0.00066842 * X * X + 0.061542 * X - 0.37089;
                else if (Dri < 175.e-6)
                    Ef_rw = 4.091e-06 * X * X * X * X - 0.00030908 * X * X * X
                          + 0.0066237 * X * X - 0.0013687 * X - 0.073022;
                else if (Dri < 250.e-6)
                    Ef_rw = 9.6719e-5 * X * X * X - 0.0068901 * X * X + 0.17305 * X - 0.65988;
                else if (Dri < 350.e-6)
                    Ef_rw = 9.0488e-5 * X * X * X - 0.006585 * X * X + 0.16606 * X - 0.56125;
                else
                    Ef_rw = 0.00010721 * X * X * X - 0.0072962 * X * X + 0.1704 * X - 0.46929;
            } else {
                vtr = -0.1021 + 4.932E3 * Dri - 0.9551E6 * Dri * Dri
                    + 0.07934E9 * Dri * Dri * Dri
                    - 0.002362E12 * Dri * Dri * Dri * Dri;
                stokes = Dcj * Dcj * vtr * rho_w / (9. * 1.718E-5 * Dri);
                reynolds = 9. * stokes / (p * p * rho_w);
                F = log(reynolds);
                G = -0.1007 - 0.358 * F + 0.0261 * F * F;
                K0 = exp(G);
                z = log(stokes / (K0 + 1.e-15));
                H = 0.1465 + 1.302 * z - 0.607 * z * z + 0.293 * z * z * z;
                yc0 = 2.0 / PI * atan(H);
                Ef_rw = (yc0 + p) * (yc0 + p) / ((1. + p) * (1. + p));
            }
            /* M:4294 runs for every cell, the zeroed ones included (Ef_rw=0);
             * SNGL is the identity in P64 */
            EFRW(o->t_Efrw, i, j) = MAXD(0.0, MIND(Ef_rw, 0.95));
        }
}

/* table_Efsw M:4307-4343 */
static void table_Efsw(th_oracle *o)
{
    for (int j = 1; j <= nbc; j++) {
        const double Dcj = o->Dc[j];
        double vtc = 1.19e4 * (1.0e4 * Dcj * Dcj * 0.25);
        for (int i = 1; i <= nbs; i++) {
            const double Dsi = o->Ds[i];
            double vts = av_s * pow(Dsi, bv_s) * exp(-fv_s * Dsi) - vtc;
            double Ds_m = pow(am_s * pow(Dsi, bm_s) / am_r, o->obmr);
            double p = Dcj / Ds_m;
            if (p > 0.25 || Dsi < D0s || Dcj < 6.E-6 || vts < 1.E-3) {
                EFSW(o->t_Efsw, i, j) = 0.0;
            } else {
                double stokes = Dcj * Dcj * vts * rho_w / (9. * 1.718E-5 * Ds_m);
                double reynolds = 9. * stokes / (p * p * rho_w);
                double F = log(reynolds);
                double G = -0.1007 - 0.358 * F + 0.0261 * F * F;
                double K0 = exp(G);
                double z = log(stokes / (K0 + 1.e-15));
                double H = 0.1465 + 1.302 * z - 0.607 * z * z + 0.293 * z * z * z;
                double yc0 = 2.0 / PI * atan(H);
                double Ef_sw = (yc0 + p) * (yc0 + p) / ((1. + p) * (1. + p));
                EFSW(o->t_Efsw, i, j) = MAXD(0.0, MIND(Ef_sw, 0.95));
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* table_dropEvap M:4400-4439: mass / number of droplets smaller than bin i for cloud water r_c(j) and number t_Nc(k).
 * Only read under is_aerosol_aware (M:2850). */
static void table_dropEvap(th_oracle *o)
{
    double N_c[nbc + 1], massc[nbc + 1];
    for (int n = 1; n <= nbc; n++) massc[n] = am_r * pow(o->Dc[n], bm_r);
    for (int k = 1; k <= nbc; k++) {
        int nu_c = NINT(1000.E6 / o->t_Nc[k]) + 2; if (nu_c > 15) nu_c = 15;
        for (int j = 1; j <= ntb_c; j++) {
            double lamc = pow(o->t_Nc[k] * am_r * o->ccg[2][nu_c] * o->ocg1[nu_c] / o->r_c[j], o->obmr);
            double N0_c = o->t_Nc[k] * o->ocg1[nu_c] * pow(lamc, o->cce[1][nu_c]);
            for (int i = 1; i <= nbc; i++) {
                N_c[i] = N0_c * th_powi(o->Dc[i], nu_c) * exp(-lamc * o->Dc[i]) * o->dtc[i];
                double summ = 0., summ2 = 0.;
                for (int n = 1; n <= i; n++) { summ = summ + massc[n] * N_c[n]; summ2 = summ2 + N_c[n]; }
                o->tpc_wev[IX3(i, j, k, nbc, ntb_c)] = summ;
                o->tnc_wev[IX3(i, j, k, nbc, ntb_c)] = summ2;
            }
        }
    }
}

/* qr_acr_qg M:3698-3833 : one (k,m) slab */
typedef struct { th_oracle *o; int km_s, km_e; const double *vr, *vg, *vs; } slab_job;

static void racg_slab(th_oracle *o, int km, const double *vr, const double *vg)
{
    const int m = km / ntb_r1 + 1, k = km % ntb_r1 + 1;
    double N_r[nbr + 1], N_g[nbg + 1];
    double lam_exp = pow(o->N0r_exp[k] * am_r * o->crg[1] / o->r_r[m], o->ore1);
    double lamr = lam_exp * pow(o->crg[3] * o->org2 * o->org1, o->obmr);
    double N0_r = o->N0r_exp[k] / (o->crg[2] * lam_exp) * pow(lamr, o->cre[2]);
    for (int n2 = 1; n2 <= nbr; n2++)
        N_r[n2] = N0_r * pow(o->Dr[n2], mu_r) * exp(-lamr * o->Dr[n2]) * o->dtr[n2];

    for (int j = 1; j <= ntb_g; j++)
        for (int i = 1; i <= ntb_g1; i++) {
            lam_exp = pow(o->N0g_exp[i] * am_g * o->cgg[1] / o->r_g[j], o->oge1);
            double lamg = lam_exp * pow(o->cgg[3] * o->ogg2 * o->ogg1, o->obmg);
            double N0_g = o->N0g_exp[i] / (o->cgg[2] * lam_exp) * pow(lamg, o->cge[2]);
            for (int n = 1; n <= nbg; n++)
                N_g[n] = N0_g * pow(o->Dg[n], mu_g) * exp(-lamg * o->Dg[n]) * o->dtg[n];

            double t1 = 0, t2 = 0, z1 = 0, z2 = 0, y1 = 0, y2 = 0;
            for (int n2 = 1; n2 <= nbr; n2++) {
                const double Drn = o->Dr[n2];
                double massr = am_r * pow(Drn, bm_r);
                for (int n = 1; n <= nbg; n++) {
                    const double Dgn = o->Dg[n];
                    double massg = am_g * pow(Dgn, bm_g);
                    double dvg = 0.5 * ((vr[n2] - vg[n]) + fabs(vr[n2] - vg[n]));
                    double dvr = 0.5 * ((vg[n] - vr[n2]) + fabs(vg[n] - vr[n2]));
                    const double c = PI * .25 * Ef_rg * (Dgn + Drn) * (Dgn + Drn);
                    t1 = t1 + c * dvg * massg * N_g[n] * N_r[n2];
                    z1 = z1 + c * dvg * massr * N_g[n] * N_r[n2];
                    y1 = y1 + c * dvg * N_g[n] * N_r[n2];
                    t2 = t2 + c * dvr * massr * N_g[n] * N_r[n2];
                    y2 = y2 + c * dvr * N_g[n] * N_r[n2];
                    z2 = z2 + c * dvr * massg * N_g[n] * N_r[n2];
                }
            }
            RACG(o->tcg_racg, i, j, k, m) = t1;
            RACG(o->tmr_racg, i, j, k, m) = MIND(z1, o->r_r[m] * 1.0);
            RACG(o->tcr_gacr, i, j, k, m) = t2;
            RACG(o->tmg_gacr, i, j, k, m) = z2;
            RACG(o->tnr_racg, i, j, k, m) = y1;
            RACG(o->tnr_gacr, i, j, k, m) = y2;
        }
}

/* Field et al. (2005) moment polynomial used at M:3939-3964 */
static double field_loga(const th_oracle *o, double tc, double x)
{
    const double *sa = o->sa;
    return sa[1] + sa[2] * tc + sa[3] * x + sa[4] * tc * x + sa[5] * tc * tc
         + sa[6] * x * x + sa[7] * tc * tc * x + sa[8] * tc * x * x
         + sa[9] * tc * tc * tc + sa[10] * x * x * x;
}
static double field_b(const th_oracle *o, double tc, double x)
{
    const double *sb = o->sb;
    return sb[1] + sb[2] * tc + sb[3] * x + sb[4] * tc * x + sb[5] * tc * tc
         + sb[6] * x * x + sb[7] * tc * tc * x + sb[8] * tc * x * x
         + sb[9] * tc * tc * tc + sb[10] * x * x * x;
}

/* qr_acr_qs M:3842-4082 : one (k,m) slab */
static void racs_slab(th_oracle *o, int km, const double *vr, const double *vs)
{
    const int m = km / ntb_r1 + 1, k = km % ntb_r1 + 1;
    double N_r[nbr + 1], N_s[nbs + 1];
    double lam_exp = pow(o->N0r_exp[k] * am_r * o->crg[1] / o->r_r[m], o->ore1);
    double lamr = lam_exp * pow(o->crg[3] * o->org2 * o->org1, o->obmr);
    double N0_r = o->N0r_exp[k] / (o->crg[2] * lam_exp) * pow(lamr, o->cre[2]);
    for (int n2 = 1; n2 <= nbr; n2++)
        N_r[n2] = N0_r * pow(o->Dr[n2], mu_r) * exp(-lamr * o->Dr[n2]) * o->dtr[n2];

    for (int j = 1; j <= ntb_t; j++)
        for (int i = 1; i <= ntb_s; i++) {
            const double Tcj = o->Tc[j];
            double M2 = o->r_s[i] * o->oams * 1.0, second, loga_, a_, b_;
            /* NB the reference takes the *polynomial* branch when bm_s==2 here
             * (M:3938), the opposite of mp_thompson's M:1553. Kept as is. */
            if (bm_s > 2.0 - 1.E-3 && bm_s < 2.0 + 1.E-3) {
                loga_ = field_loga(o, Tcj, bm_s);
                a_ = pow(10.0, loga_);
                b_ = field_b(o, Tcj, bm_s);
                second = pow(M2 / a_, 1. / b_);
            } else {
                second = M2;
            }
            loga_ = field_loga(o, Tcj, o->cse[1]);
            a_ = pow(10.0, loga_);
            b_ = field_b(o, Tcj, o->cse[1]);
            double M3 = a_ * pow(second, b_);
            double oM3 = 1. / M3;
            double Mrat = M2 * (M2 * oM3) * (M2 * oM3) * (M2 * oM3);
            double M0 = pow(M2 * oM3, mu_s);
            double slam1 = M2 * oM3 * Lam0;
            double slam2 = M2 * oM3 * Lam1;
            for (int n = 1; n <= nbs; n++)
                N_s[n] = Mrat * (Kap0 * exp(-slam1 * o->Ds[n])
                       + Kap1 * M0 * pow(o->Ds[n], mu_s) * exp(-slam2 * o->Ds[n])) * o->dts[n];

            double t1 = 0, t2 = 0, t3 = 0, t4 = 0, z1 = 0, z2 = 0, z3 = 0, z4 = 0,
                   y1 = 0, y2 = 0, y3 = 0, y4 = 0;
            for (int n2 = 1; n2 <= nbr; n2++) {
                const double Drn = o->Dr[n2];
                double massr = am_r * pow(Drn, bm_r);
                for (int n = 1; n <= nbs; n++) {
                    const double Dsn = o->Ds[n];
                    double masss = am_s * pow(Dsn, bm_s);
                    double dvs = 0.5 * ((vr[n2] - vs[n]) + fabs(vr[n2] - vs[n]));
                    double dvr = 0.5 * ((vs[n] - vr[n2]) + fabs(vs[n] - vr[n2]));
                    const double c = PI * .25 * Ef_rs * (Dsn + Drn) * (Dsn + Drn);
                    if (massr > 1.5 * masss) {
                        t1 = t1 + c * dvs * masss * N_s[n] * N_r[n2];
                        z1 = z1 + c * dvs * massr * N_s[n] * N_r[n2];
                        y1 = y1 + c * dvs * N_s[n] * N_r[n2];
                    } else {
                        t3 = t3 + c * dvs * masss * N_s[n] * N_r[n2];
                        z3 = z3 + c * dvs * massr * N_s[n] * N_r[n2];
                        y3 = y3 + c * dvs * N_s[n] * N_r[n2];
                    }
                    if (massr > 1.5 * masss) {
                        t2 = t2 + c * dvr * massr * N_s[n] * N_r[n2];
                        y2 = y2 + c * dvr * N_s[n] * N_r[n2];
                        z2 = z2 + c * dvr * masss * N_s[n] * N_r[n2];
                    } else {
                        t4 = t4 + c * dvr * massr * N_s[n] * N_r[n2];
                        y4 = y4 + c * dvr * N_s[n] * N_r[n2];
                        z4 = z4 + c * dvr * masss * N_s[n] * N_r[n2];
                    }
                }
            }
            RACS(o->tcs_racs1, i, j, k, m) = t1;
            RACS(o->tmr_racs1, i, j, k, m) = MIND(z1, o->r_r[m] * 1.0);
            RACS(o->tcs_racs2, i, j, k, m) = t3;
            RACS(o->tmr_racs2, i, j, k, m) = z3;
            RACS(o->tcr_sacr1, i, j, k, m) = t2;
            RACS(o->tms_sacr1, i, j, k, m) = z2;
            RACS(o->tcr_sacr2, i, j, k, m) = t4;
            RACS(o->tms_sacr2, i, j, k, m) = z4;
            RACS(o->tnr_racs1, i, j, k, m) = y1;
            RACS(o->tnr_racs2, i, j, k, m) = y3;
            RACS(o->tnr_sacr1, i, j, k, m) = y2;
            RACS(o->tnr_sacr2, i, j, k, m) = y4;
        }
}

static void *slab_worker(void *arg)
{
    slab_job *jb = (slab_job *)arg;
    for (int km = jb->km_s; km <= jb->km_e; km++) {
        if (jb->vg) racg_slab(jb->o, km, jb->vr, jb->vg);
        else        racs_slab(jb->o, km, jb->vr, jb->vs);
    }
    return NULL;
}

/* contiguous zero-based [km_s,km_e] split of the ntb_r*ntb_r1 slabs over
 * threads: what wrf_dm_decomp1d would hand each rank (M:3745, M:3913) */
static void run_slabs(th_oracle *o, const double *vr, const double *vg, const double *vs)
{
    int nt = o->nthreads < 1 ? 1 : o->nthreads;
    const int total = ntb_r * ntb_r1;
    if (nt > total) nt = total;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
    slab_job *jb = (slab_job *)malloc(sizeof(slab_job) * nt);
    for (int t = 0; t < nt; t++) {
        jb[t].o = o; jb[t].vr = vr; jb[t].vg = vg; jb[t].vs = vs;
        jb[t].km_s = (int)((long)total * t / nt);
        jb[t].km_e = (int)((long)total * (t + 1) / nt) - 1;
        if (nt == 1) slab_worker(&jb[t]);
        else pthread_create(&th[t], NULL, slab_worker, &jb[t]);
    }
    if (nt > 1) for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
    free(th); free(jb);
}

static void qr_acr_qg(th_oracle *o)
{
    double vr[nbr + 1], vg[nbg + 1];
    for (int n2 = 1; n2 <= nbr; n2++) {                      /* M:3731-3736 */
        const double D = o->Dr[n2];
        vr[n2] = -0.1021 + 4.932E3 * D - 0.9551E6 * D * D
               + 0.07934E9 * D * D * D - 0.002362E12 * D * D * D * D;
    }
    for (int n = 1; n <= nbg; n++) vg[n] = av_g * pow(o->Dg[n], bv_g);   /* M:3738 */
    run_slabs(o, vr, vg, NULL);
}

static void qr_acr_qs(th_oracle *o)
{
    double vr[nbr + 1], vs[nbs + 1];
    for (int n2 = 1; n2 <= nbr; n2++) {                      /* M:3898-3904 */
        const double D = o->Dr[n2];
        vr[n2] = -0.1021 + 4.932E3 * D - 0.9551E6 * D * D
               + 0.07934E9 * D * D * D - 0.002362E12 * D * D * D * D;
        /* D1(n2) at M:3903 is computed and never used */
    }
    for (int n = 1; n <= nbs; n++)
        vs[n] = 1.5 * av_s * pow(o->Ds[n], bv_s) * exp(-fv_s * o->Ds[n]);   /* M:3906 */
    run_slabs(o, vr, NULL, vs);
}

/* freezeH2O M:4092-4175.  The outer m=1..ntb_IN loop (M:4118) rewrites the
 * same cells on every pass, so only m=ntb_IN (T_adjust from Nt_IN(55))
 * survives; that single pass is what is evaluated here. */
static void freezeH2O(th_oracle *o)
{
    double massr[nbr + 1], massc[nbc + 1], N_r[nbr + 1], N_c[nbc + 1];
    const double orho_w = 1. / rho_w;
    for (int n2 = 1; n2 <= nbr; n2++) massr[n2] = am_r * pow(o->Dr[n2], bm_r);
    for (int n = 1; n <= nbc; n++)   massc[n] = am_r * pow(o->Dc[n], bm_r);

    const int m = ntb_IN;
    const double T_adjust = MAXD(-3.0, MIND(3.0 - log10(o->Nt_IN[m]), 3.0));
    for (int k = 1; k <= 45; k++) {
        const double Texp = exp((double)k - T_adjust * 1.0) - 1.0;
        for (int j = 1; j <= ntb_r1; j++)
            for (int i = 1; i <= ntb_r; i++) {
                double lam_exp = pow(o->N0r_exp[j] * am_r * o->crg[1] / o->r_r[i], o->ore1);
                double lamr = lam_exp * pow(o->crg[3] * o->org2 * o->org1, o->obmr);
                double N0_r = o->N0r_exp[j] / (o->crg[2] * lam_exp) * pow(lamr, o->cre[2]);
                double sum1 = 0, sum2 = 0, sumn1 = 0, sumn2 = 0;
                for (int n2 = nbr; n2 >= 1; n2--) {
                    N_r[n2] = N0_r * pow(o->Dr[n2], mu_r) * exp(-lamr * o->Dr[n2]) * o->dtr[n2];
                    double vol = massr[n2] * orho_w;
                    double prob = 1.0 - exp(-120.0 * vol * 5.2e-4 * Texp);
                    if (massr[n2] < o->xm0g) {
                        sumn1 = sumn1 + prob * N_r[n2];
                        sum1 = sum1 + prob * N_r[n2] * massr[n2];
                    } else {
                        sumn2 = sumn2 + prob * N_r[n2];
                        sum2 = sum2 + prob * N_r[n2] * massr[n2];
                    }
                }
                QRFZ(o->tpi_qrfz, i, j, k) = sum1;
                QRFZ(o->tni_qrfz, i, j, k) = sumn1;
                QRFZ(o->tpg_qrfz, i, j, k) = sum2;
                QRFZ(o->tnr_qrfz, i, j, k) = sumn2;
            }

        int nu_c = NINT(1000.E6 / o->t_Nc[1]) + 2;          /* M:4155 */
        if (nu_c > 15) nu_c = 15;
        for (int i = 1; i <= ntb_c; i++) {
            double lamc = pow(o->t_Nc[1] * am_r * o->ccg[2][nu_c] * o->ocg1[nu_c] / o->r_c[i], o->obmr);
            double N0_c = o->t_Nc[1] * o->ocg1[nu_c] * pow(lamc, o->cce[1][nu_c]);
            double sum1 = 0, sumn2 = 0;
            for (int n = nbc; n >= 1; n--) {
                double vol = massc[n] * orho_w;
                double prob = 1.0 - exp(-120.0 * vol * 5.2e-4 * Texp);
                N_c[n] = N0_c * th_powi(o->Dc[n], nu_c) * exp(-lamc * o->Dc[n]) * o->dtc[n];
                sumn2 = MIND(o->t_Nc[1], sumn2 + prob * N_c[n]);
                sum1 = sum1 + prob * N_c[n] * massc[n];
                if (sum1 >= o->r_c[i]) break;
            }
            QCFZ(o->tpi_qcfz, i, k) = sum1;
            QCFZ(o->tni_qcfz, i, k) = sumn2;
        }
    }
}

/* qi_aut_qs M:4190-4233 */
static void qi_aut_qs(th_oracle *o)
{
    double N_i[nbi + 1];
    for (int j = 1; j <= ntb_i1; j++)
        for (int i = 1; i <= ntb_i; i++) {
            double lami = pow(am_i * o->cig[2] * o->oig1 * o->Nt_i[j] / o->r_i[i], o->obmi);
            double Di_mean = (bm_i + mu_i + 1.) / lami;
            double N0_i = o->Nt_i[j] * o->oig1 * pow(lami, o->cie[1]);
            double t1 = 0, t2 = 0;
            if (Di_mean > 5. * D0s) {
                t1 = o->r_i[i];
                t2 = o->Nt_i[j];
                IAUS(o->tpi_ide, i, j) = 0.0;
            } else if (Di_mean < o->D0i) {
                t1 = 0; t2 = 0;
                IAUS(o->tpi_ide, i, j) = 1.0;
            } else {
                double xlimit_intg = lami * D0s;
                IAUS(o->tpi_ide, i, j) = th_oracle_gammp(mu_i + 2.0, xlimit_intg) * 1.0;
                for (int n2 = 1; n2 <= nbi; n2++) {
                    N_i[n2] = N0_i * pow(o->Di[n2], mu_i) * exp(-lami * o->Di[n2]) * o->dti[n2];
                    if (o->Di[n2] >= D0s) {
                        t1 = t1 + N_i[n2] * am_i * pow(o->Di[n2], bm_i);
                        t2 = t2 + N_i[n2];
                    }
                }
            }
            IAUS(o->tps_iaus, i, j) = t1;
            IAUS(o->tni_iaus, i, j) = t2;
        }
}

/* ------------------------------------------------------------------ */
/* private binary cache of the 4-D tables (tens of seconds to rebuild) */
#define CACHE_MAGIC 0x54483039u
typedef struct { const char *name; double **p; size_t n; } tabent;

static int table_list(th_oracle *o, tabent *t)
{
    const size_t ng = (size_t)ntb_g1 * ntb_g * ntb_r1 * ntb_r;
    const size_t ns = (size_t)ntb_s * ntb_t * ntb_r1 * ntb_r;
    const size_t nz3 = (size_t)ntb_r * ntb_r1 * 45, nc2 = (size_t)ntb_c * 45;
    const size_t ni2 = (size_t)ntb_i * ntb_i1;
    int n = 0;
#define T(nm, sz) t[n].name = #nm; t[n].p = &o->nm; t[n].n = sz; n++;
    T(tcg_racg, ng) T(tmr_racg, ng) T(tcr_gacr, ng) T(tmg_gacr, ng) T(tnr_racg, ng) T(tnr_gacr, ng)
    T(tcs_racs1, ns) T(tmr_racs1, ns) T(tcs_racs2, ns) T(tmr_racs2, ns) T(tcr_sacr1, ns)
    T(tms_sacr1, ns) T(tcr_sacr2, ns) T(tms_sacr2, ns) T(tnr_racs1, ns) T(tnr_racs2, ns)
    T(tnr_sacr1, ns) T(tnr_sacr2, ns)
    T(tpi_qcfz, nc2) T(tni_qcfz, nc2)
    T(tpi_qrfz, nz3) T(tpg_qrfz, nz3) T(tni_qrfz, nz3) T(tnr_qrfz, nz3)
    T(tps_iaus, ni2) T(tni_iaus, ni2) T(tpi_ide, ni2)
    T(t_Efrw, (size_t)nbr * nbc) T(t_Efsw, (size_t)nbs * nbc)
#undef T
    return n;
}

static int cache_load(th_oracle *o, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    unsigned hdr[2]; double nc;
    tabent t[40]; int n = table_list(o, t), ok = 1;
    if (fread(hdr, sizeof hdr, 1, f) != 1 || hdr[0] != CACHE_MAGIC || hdr[1] != (unsigned)n) ok = 0;
    if (ok && (fread(&nc, sizeof nc, 1, f) != 1 || nc != o->set_Nc)) ok = 0;
    for (int i = 0; ok && i < n - 2; i++)          /* Ef tables are always rebuilt */
        if (fread(*t[i].p, sizeof(double), t[i].n, f) != t[i].n) ok = 0;
    fclose(f);
    return ok;
}
static void cache_store(th_oracle *o, const char *path)
{
    char tmp[4096];
    snprintf(tmp, sizeof tmp, "%s.tmp.%ld", path, (long)getpid());
    FILE *f = fopen(tmp, "wb");
    if (!f) return;
    tabent t[40]; int n = table_list(o, t);
    unsigned hdr[2] = { CACHE_MAGIC, (unsigned)n };
    fwrite(hdr, sizeof hdr, 1, f);
    fwrite(&o->set_Nc, sizeof(double), 1, f);
    for (int i = 0; i < n - 2; i++) fwrite(*t[i].p, sizeof(double), t[i].n, f);
    fclose(f);
    rename(tmp, path);
}

/* ------------------------------------------------------------------ */
th_oracle *th_oracle_create(int iiwarm, double set_Nc, int l_sediment,
                            int nthreads, const char *cache_path)
{
    th_oracle *o = (th_oracle *)calloc(1, sizeof *o);
    if (!o) return NULL;
    o->iiwarm = iiwarm; o->l_sediment = l_sediment; o->set_Nc = set_Nc;
    o->nthreads = nthreads;
    o->Nt_c = set_Nc * 1.e6;                                  /* M:381 */

    /* axes M:215-315 */
    fill_decades(o->r_c, ntb_c, 1.e-6);
    fill_decades(o->r_i, ntb_i, 1.e-10);
    fill_decades(o->r_r, ntb_r, 1.e-6);
    fill_decades(o->r_g, ntb_g, 1.e-5);
    fill_decades(o->r_s, ntb_s, 1.e-5);
    fill_decades(o->N0r_exp, ntb_r1, 1.e6);
    fill_decades(o->N0g_exp, ntb_g1, 1.e4);
    fill_decades(o->Nt_i, ntb_i1, 1.0);
    fill_decades(o->Nt_IN, ntb_IN, 1.0);
    {
        static const double sa_[10] = { 5.065339, -0.062659, -3.032362, 0.029469, -0.000285,
                                        0.31255, 0.000204, 0.003199, 0.0, -0.015952 };
        static const double sb_[10] = { 0.476221, -0.015896, 0.165977, 0.007468, -0.000141,
                                        0.060366, 0.000079, 0.000594, 0.0, -0.003577 };
        static const double Tc_[9] = { -0.01, -5., -10., -15., -20., -25., -30., -35., -40. };
        for (int i = 0; i < 10; i++) { o->sa[i + 1] = sa_[i]; o->sb[i + 1] = sb_[i]; }
        for (int i = 0; i < 9; i++) o->Tc[i + 1] = Tc_[i];
    }

    /* M:442-447 */
    o->Sc3 = pow(Sc, 1. / 3.);
    o->D0i = pow(xm0i / am_i, 1. / bm_i);
    o->xm0s = am_s * pow(D0s, bm_s);
    o->xm0g = am_g * pow(D0g, bm_g);

    /* M:452-465 */
    for (int n = 1; n <= 15; n++) {
        o->cce[1][n] = n + 1.;
        o->cce[2][n] = bm_r + n + 1.;
        o->cce[3][n] = bm_r + n + 4.;
        o->cce[4][n] = n + bv_c + 1.;
        o->cce[5][n] = bm_r + n + bv_c + 1.;
        for (int i = 1; i <= 5; i++) o->ccg[i][n] = wgamma(o->cce[i][n]);
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
    for (int n = 1; n <= 7; n++) o->cig[n] = wgamma(o->cie[n]);
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
    for (int n = 1; n <= 13; n++) o->crg[n] = wgamma(o->cre[n]);
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
    for (int n = 1; n <= 18; n++) o->csg[n] = wgamma(o->cse[n]);
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
    for (int n = 1; n <= 12; n++) o->cgg[n] = wgamma(o->cge[n]);
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
    o->t2_qr_ev = 0.308 * o->Sc3 * sqrt(av_r) * o->crg[11];
    o->t1_qs_sd = 0.86;
    o->t2_qs_sd = 0.28 * o->Sc3 * sqrt(av_s);
    o->t1_qs_me = PI * 4. * C_sqrd * olfus * 0.86;
    o->t2_qs_me = PI * 4. * C_sqrd * olfus * 0.28 * o->Sc3 * sqrt(av_s);
    o->t1_qg_sd = 0.86 * o->cgg[10];
    o->t2_qg_sd = 0.28 * o->Sc3 * sqrt(av_g) * o->cgg[11];
    o->t1_qg_me = PI * 4. * C_cube * olfus * 0.86 * o->cgg[10];
    o->t2_qg_me = PI * 4. * C_cube * olfus * 0.28 * o->Sc3 * sqrt(av_g) * o->cgg[11];

    /* M:594-602 */
    o->nic2 = NINT(log10(o->r_c[1]));
    o->nii2 = NINT(log10(o->r_i[1]));
    o->nii3 = NINT(log10(o->Nt_i[1]));
    o->nir2 = NINT(log10(o->r_r[1]));
    o->nir3 = NINT(log10(o->N0r_exp[1]));
    o->nis2 = NINT(log10(o->r_s[1]));
    o->nig2 = NINT(log10(o->r_g[1]));
    o->nig3 = NINT(log10(o->N0g_exp[1]));
    o->niIN2 = NINT(log10(o->Nt_IN[1]));

    /* bins M:605-669 */
    o->Dc[1] = D0c * 1.0; o->dtc[1] = D0c * 1.0;
    for (int n = 2; n <= nbc; n++) {
        o->Dc[n] = o->Dc[n - 1] + 1.0e-6;
        o->dtc[n] = (o->Dc[n] - o->Dc[n - 1]);
    }
    make_bins(o->D0i * 1.0, 5.0 * D0s, nbi, o->Di, o->dti);
    make_bins(D0r * 1.0, 0.005, nbr, o->Dr, o->dtr);
    make_bins(D0s * 1.0, 0.02, nbs, o->Ds, o->dts);
    make_bins(D0g * 1.0, 0.05, nbg, o->Dg, o->dtg);
    make_bins(1.0, 3000.0, nbc, o->t_Nc, NULL);
    for (int n = 1; n <= nbc; n++) o->t_Nc[n] = o->t_Nc[n] * 1.e6;
    o->nic1 = (int)log(o->t_Nc[nbc] / o->t_Nc[1]);   /* INTEGER nic1 truncates, M:195,670 (U3) */

    /* tables M:386-423 */
    tabent t[40]; int nt = table_list(o, t);
    for (int i = 0; i < nt; i++) *t[i].p = talloc(t[i].n);

    table_Efrw(o);                                            /* M:766 */
    table_Efsw(o);                                            /* M:767 */
    o->tnc_wev = talloc((size_t)nbc * ntb_c * nbc);
    o->tpc_wev = talloc((size_t)nbc * ntb_c * nbc);
    table_dropEvap(o);                                        /* M:771 */
    if (!iiwarm) {                                            /* M:773-791 */
        if (!(cache_path && cache_load(o, cache_path))) {
            qr_acr_qg(o);
            qr_acr_qs(o);
            freezeH2O(o);
            qi_aut_qs(o);
            if (cache_path) cache_store(o, cache_path);
        }
    }
    o->view = th_oracle_make_view(o);
    o->view_p32n = th_oracle_make_view_p32n(o);
    if (!o->view || !o->view_p32n) { th_oracle_destroy(o); return NULL; }
    return o;
}

void th_oracle_destroy(th_oracle *o)
{
    if (!o) return;
    tabent t[40]; int nt = table_list(o, t);
    for (int i = 0; i < nt; i++) free(*t[i].p);
    free(o->view); free(o->view_p32n); free(o->tnc_wev); free(o->tpc_wev);
    free(o);
}

void th_oracle_set_aerosol_aware(th_oracle *o, int flag)
{
    o->is_aerosol_aware = flag != 0;
    free(o->view); free(o->view_p32n);
    o->view = th_oracle_make_view(o);
    o->view_p32n = th_oracle_make_view_p32n(o);
}

/* ------------------------------------------------------------------ */
const double *th_oracle_table(const th_oracle *oc, const char *name, int *ndim, int dims[4])
{
    th_oracle *o = (th_oracle *)oc;
    if (!strcmp(name, "tnc_wev")) { *ndim = 3; dims[0] = nbc; dims[1] = ntb_c; dims[2] = nbc; return o->tnc_wev; }
    tabent t[40]; int nt = table_list(o, t);
    for (int i = 0; i < nt; i++)
        if (!strcmp(name, t[i].name)) {
            if (i < 6)       { *ndim = 4; dims[0] = ntb_g1; dims[1] = ntb_g; dims[2] = ntb_r1; dims[3] = ntb_r; }
            else if (i < 18) { *ndim = 4; dims[0] = ntb_s; dims[1] = ntb_t; dims[2] = ntb_r1; dims[3] = ntb_r; }
            else if (i < 20) { *ndim = 2; dims[0] = ntb_c; dims[1] = 45; }
            else if (i < 24) { *ndim = 3; dims[0] = ntb_r; dims[1] = ntb_r1; dims[2] = 45; }
            else if (i < 27) { *ndim = 2; dims[0] = ntb_i; dims[1] = ntb_i1; }
            else             { *ndim = 2; dims[0] = 100; dims[1] = 100; }
            return *t[i].p;
        }
    return NULL;
}

const double *th_oracle_const(const th_oracle *o, const char *name, int *n)
{
#define C1(nm)      if (!strcmp(name, #nm)) { *n = 1; return &o->nm; }
#define CA(nm, len) if (!strcmp(name, #nm)) { *n = len; return &o->nm[1]; }
    C1(Nt_c) C1(Sc3) C1(D0i) C1(xm0s) C1(xm0g)
    C1(oig1) C1(oig2) C1(obmi) C1(ore1) C1(org1) C1(org2) C1(org3) C1(obmr)
    C1(oams) C1(obms) C1(ocms) C1(oge1) C1(ogg1) C1(ogg2) C1(ogg3) C1(oamg) C1(obmg) C1(ocmg)
    C1(t1_qr_qc) C1(t1_qr_qi) C1(t2_qr_qi) C1(t1_qg_qc) C1(t1_qs_qc) C1(t1_qs_qi)
    C1(t1_qr_ev) C1(t2_qr_ev) C1(t1_qs_sd) C1(t2_qs_sd) C1(t1_qg_sd) C1(t2_qg_sd)
    C1(t1_qs_me) C1(t2_qs_me) C1(t1_qg_me) C1(t2_qg_me)
    CA(cie, 7) CA(cig, 7) CA(cre, 13) CA(crg, 13) CA(cse, 18) CA(csg, 18) CA(cge, 12) CA(cgg, 12)
    CA(ocg1, 15) CA(ocg2, 15)
    CA(Dc, nbc) CA(dtc, nbc) CA(Di, nbi) CA(dti, nbi) CA(Dr, nbr) CA(dtr, nbr)
    CA(Ds, nbs) CA(dts, nbs) CA(Dg, nbg) CA(dtg, nbg) CA(t_Nc, nbc)
    CA(r_c, ntb_c) CA(r_i, ntb_i) CA(r_r, ntb_r) CA(r_g, ntb_g) CA(r_s, ntb_s)
    CA(N0r_exp, ntb_r1) CA(N0g_exp, ntb_g1) CA(Nt_i, ntb_i1)
    /* cce/ccg rows: "cce1".."cce5", "ccg1".."ccg5" (15 entries each) */
    if ((!strncmp(name, "cce", 3) || !strncmp(name, "ccg", 3)) && name[3] >= '1' && name[3] <= '5' && !name[4]) {
        *n = 15;
        return name[2] == 'e' ? &o->cce[name[3] - '0'][1] : &o->ccg[name[3] - '0'][1];
    }
#undef C1
#undef CA
    *n = 0;
    return NULL;
}

int th_oracle_int(const th_oracle *o, const char *name)
{
#define CI(nm) if (!strcmp(name, #nm)) return o->nm;
    CI(nic1) CI(nic2) CI(nii2) CI(nii3) CI(nir2) CI(nir3) CI(nis2) CI(nig2) CI(nig3) CI(niIN2)
    CI(iiwarm) CI(l_sediment)
#undef CI
    return -999999;
}
