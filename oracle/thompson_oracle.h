/*
 * thompson_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the Thompson-09n column microphysics that the KiD
 * driver calls.  Citation legend (paths under /root/reference):
 *     M: = module_mp_thompson09n.f90      W: = mphys_thompson09n.f90
 *
 * Arithmetic model: "P64" of SURVEY.md section 8c -- every REAL and DOUBLE
 * PRECISION is IEEE binary64 and every literal is the binary64 nearest to its
 * decimal text (what `flang -fdefault-real-8 -fdefault-double-8` produces).
 * The *_p32n entry points run the same source in the reference's native "P32n"
 * arithmetic (REAL = binary32), for the config-5 precision sweep.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path (kid_amd/) never links or calls it.
 *
 * PARITY PINNING: the reference cannot be built in this image under the rules
 * of this project (it USEs five KiD modules that are not in the mount and
 * needs a source patch for undefined behaviour U1), and it ships no golden
 * vectors.  The oracle is pinned against the known-answer values that the
 * survey recorded from the reference itself (KAT-A/B/C,
 * 6-7 significant digits); see tests/test_oracle_kat.py.  Beyond those
 * digits parity is UNPINNED.
 */
#ifndef THOMPSON_ORACLE_H
#define THOMPSON_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct th_oracle th_oracle;

/* number of per-level process-rate diagnostics, emission order of M:2967-3119 */
#define TH_ORACLE_NRATES 36

/* thompson_init (M:374-797).  iiwarm/set_Nc come from KiD `namelists`
 * (M:22), l_sediment from `switches` (M:20).  nthreads parallelises the
 * (k,m) slabs of qr_acr_qg / qr_acr_qs exactly like the compiled-out
 * wrf_dm_decomp1d split (M:3744-3753, M:3912-3921).  cache_path (may be NULL)
 * is a private binary cache of the computed tables (not the reference's
 * run_data text format). */
th_oracle *th_oracle_create(int iiwarm, double set_Nc, int l_sediment,
                            int nthreads, const char *cache_path);
void th_oracle_destroy(th_oracle *o);
/* is_aerosol_aware (M:28; .false. in KiD and by default here): .true. switches on the aerosol-aware branch --
 * prognostic droplet number, activ_ncloud, iceDeMott, iceKoop, aerosol scavenging tendencies, droplet evaporation
 * from tnc_wev (M:1410, M:2043-2111, M:2397-2408, M:2602, M:2796-2852, M:2867).  Not thread-safe against running steps. */
void th_oracle_set_aerosol_aware(th_oracle *o, int flag);

/* mp_thompson (M:1156-3688): one column, one step.  Arrays hold levels
 * kts..kte as C index 0..nz-1.  ppt = {pptrain, pptsnow, pptgraul, pptice}
 * (INOUT, accumulated).  rates (may be NULL) receives
 * [TH_ORACLE_NRATES][nz] values in the save_dg order of M:2967-3119 (all
 * zero when the no_micro early return at M:1540 is taken).
 * nstep_out (may be NULL) receives the 4 substep counts (rain, ice, snow,
 * graupel) of M:3365,3447,3504,3553.  Returns 0, or 1 if no_micro. */
int th_oracle_mp_thompson(const th_oracle *o,
                          double *qv1d, double *qc1d, double *qi1d,
                          double *qr1d, double *qs1d, double *qg1d,
                          double *ni1d, double *nr1d, double *nc1d,
                          double *nwfa1d, double *nifa1d, double *t1d,
                          const double *p1d, const double *w1d,
                          const double *dzq, double ppt[4],
                          int nz, double dt, double *rates, int *nstep_out);

/* Same, plus conditioning diagnostics: illcond (may be NULL) receives one int
 * per level, non-zero where a branch of the reference was decided on a
 * cancellation residue (bit 0: `xri > 0.` at M:3587, bit 1: `xrc > 0.` at
 * M:3596 after the species was removed completely).  At such levels the
 * reference's own output is chaotic (a 1-ulp change upstream flips it), so
 * parity tests compare them against both outcomes (th_oracle_mp_thompson_force). */
int th_oracle_mp_thompson_ex(const th_oracle *o,
                             double *qv1d, double *qc1d, double *qi1d,
                             double *qr1d, double *qs1d, double *qg1d,
                             double *ni1d, double *nr1d, double *nc1d,
                             double *nwfa1d, double *nifa1d, double *t1d,
                             const double *p1d, const double *w1d,
                             const double *dzq, double ppt[4],
                             int nz, double dt, double *rates, int *nstep_out,
                             int *illcond);

/* Same, with the two residue-decided tests of block Q forced (oracle only; 0 = the reference's own decision,
 * 1 = taken, 2 = not taken, applied only at the levels th_oracle_mp_thompson_ex flags): the two outcomes an
 * implementation with different rounding may legitimately produce there.  Parity tests require every flagged
 * level of the HIP result to equal one of them (tests/parity.py). */
int th_oracle_mp_thompson_force(const th_oracle *o,
                                double *qv1d, double *qc1d, double *qi1d,
                                double *qr1d, double *qs1d, double *qg1d,
                                double *ni1d, double *nr1d, double *nc1d,
                                double *nwfa1d, double *nifa1d, double *t1d,
                                const double *p1d, const double *w1d,
                                const double *dzq, double ppt[4],
                                int nz, double dt, double *rates, int *nstep_out,
                                int *illcond, int force);

/* Batch of columns, k-fastest layout x[col*nz + k]; ppt[col*4 + s].
 * Runs th_oracle_mp_thompson per column on nthreads host threads
 * (cpu_baseline leg of bench.py). */
int th_oracle_batch(const th_oracle *o, long ncol, int nz, double dt,
                    double *qv, double *qc, double *qi, double *qr,
                    double *qs, double *qg, double *ni, double *nr,
                    double *nc, double *nwfa, double *nifa, double *t,
                    const double *p, const double *w, const double *dz,
                    double *ppt, int nthreads);

int th_oracle_batch_ex(const th_oracle *o, long ncol, int nz, double dt,
                       double *qv, double *qc, double *qi, double *qr,
                       double *qs, double *qg, double *ni, double *nr,
                       double *nc, double *nwfa, double *nifa, double *t,
                       const double *p, const double *w, const double *dz,
                       double *ppt, int nthreads, int *illcond /* [ncol*nz] or NULL */);

int th_oracle_batch_force(const th_oracle *o, long ncol, int nz, double dt,
                          double *qv, double *qc, double *qi, double *qr,
                          double *qs, double *qg, double *ni, double *nr,
                          double *nc, double *nwfa, double *nifa, double *t,
                          const double *p, const double *w, const double *dz,
                          double *ppt, int nthreads, int *illcond /* [ncol*nz] or NULL */, int force);

/* ---- P32n: the reference's native arithmetic (REAL = binary32, DOUBLE PRECISION = binary64; see the header of
 * thompson_oracle_column.c).  Same entry points on float arrays; rates stay double (they are DOUBLE PRECISION). */
int th_oracle_mp_thompson_force_p32n(const th_oracle *o,
                                     float *qv1d, float *qc1d, float *qi1d, float *qr1d, float *qs1d, float *qg1d,
                                     float *ni1d, float *nr1d, float *nc1d, float *nwfa1d, float *nifa1d, float *t1d,
                                     const float *p1d, const float *w1d, const float *dzq, float ppt[4],
                                     int nz, float dt, double *rates, int *nstep_out, int *illcond, int force);
int th_oracle_batch_force_p32n(const th_oracle *o, long ncol, int nz, float dt,
                               float *qv, float *qc, float *qi, float *qr, float *qs, float *qg, float *ni, float *nr,
                               float *nc, float *nwfa, float *nifa, float *t,
                               const float *p, const float *w, const float *dz,
                               float *ppt, int nthreads, int *illcond, int force);
void th_oracle_default_aerosols_p32n(const th_oracle *o, int nz, const float *qv1d, const float *t1d, const float *p1d,
                                     float *nc1d, float *nwfa1d, float *nifa1d);
int th_oracle_kid_interface_p32n(const th_oracle *o, int nz, int nx, float dt, float p0, float r_on_cp,
                                 const float *theta, const float *dtheta_adv, const float *dtheta_div,
                                 const float *exner, const float *dz, const float *qv, const float *dqv_adv,
                                 const float *dqv_div, const float *hydro, const float *dhydro_adv,
                                 const float *dhydro_div, float *dtheta_mphys, float *dqv_mphys,
                                 float *dhydro_mphys, float *ppt);
/* the constants mp_thompson reads in each arithmetic, by name (tests): -1e30 if unknown */
double th_oracle_view_const(const th_oracle *o, const char *name, int idx);
double th_oracle_view_const_p32n(const th_oracle *o, const char *name, int idx);

/* calc_effectRad (M:4834-4935): effective radii of cloud water / cloud ice / snow for radiation coupling; re_* are
 * INOUT (levels without the species keep the caller's value). */
void th_oracle_calc_effectRad(const th_oracle *o, int nz,
                              const double *t1d, const double *p1d, const double *qv1d, const double *qc1d,
                              const double *nc1d, const double *qi1d, const double *ni1d, const double *qs1d,
                              double *re_qc1d, double *re_qi1d, double *re_qs1d);

/* Non-aerosol defaults for the inputs the KiD wrapper leaves unset
 * (decision U2 of SURVEY 8c; formulas of M:958-964). */
void th_oracle_default_aerosols(const th_oracle *o, int nz,
                                const double *qv1d, const double *t1d,
                                const double *p1d, double *nc1d,
                                double *nwfa1d, double *nifa1d);

/* mphys_thompson09_interfacen (W:28-310) for nx columns: gathers
 * profiles from KiD-style state (k fastest: a[k + nz*i]; hydrometeor
 * arrays a[k + nz*(i + nx*(ih + 5*imom))], ih=0..4 cloud,rain,ice,snow,
 * graupel, imom=0 mass, 1 number), calls mp_thompson, backs out the
 * *_mphys tendencies (W:198-245) and returns ppt[4][nx] (rain, snow,
 * graupel, ice).  nc1d/nwfa1d/nifa1d/w1d get the U2 defaults. */
int th_oracle_kid_interface(const th_oracle *o, int nz, int nx, double dt,
                            double p0, double r_on_cp,
                            const double *theta, const double *dtheta_adv,
                            const double *dtheta_div, const double *exner,
                            const double *dz, const double *qv,
                            const double *dqv_adv, const double *dqv_div,
                            const double *hydro, const double *dhydro_adv,
                            const double *dhydro_div,
                            double *dtheta_mphys, double *dqv_mphys,
                            double *dhydro_mphys, double *ppt);

/* Introspection for tests: scalar/array constants and lookup tables.
 * th_oracle_const returns a pointer to n doubles (NULL if unknown).
 * th_oracle_table returns column-major (Fortran-order) data + dims. */
const double *th_oracle_const(const th_oracle *o, const char *name, int *n);
const double *th_oracle_table(const th_oracle *o, const char *name,
                              int *ndim, int dims[4]);
int th_oracle_int(const th_oracle *o, const char *name);

/* scalar helpers exposed for unit tests */
double th_oracle_rslf(double p, double t);     /* M:4656-4686 */
double th_oracle_rsif(double p, double t);     /* M:4691-4717 */
double th_oracle_gammln(double xx);            /* M:4598-4620 */
double th_oracle_gammp(double a, double x);    /* M:4623-4641 */

#ifdef __cplusplus
}
#endif
#endif
