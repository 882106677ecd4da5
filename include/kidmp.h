/*
 * kidmp.h -- C ABI of the MI355X-native Thompson-09n column microphysics.
 *
 * This is the drop-in boundary for the one hot path of EnverRamirez/KiD that
 * this project replaces.  Reference files (read-only mount /root/reference):
 *     M: = module_mp_thompson09n.f90        W: = mphys_thompson09n.f90
 * Every entry point names the reference interface it replaces.  Signatures use
 * plain pointers and sizes only; the Fortran side binds them through
 * ISO_C_BINDING (kid_amd/fortran/module_mp_thompson09n.f90).
 *
 * Conventions
 *   - The kidmp_* entries take IEEE binary64 arrays and compute in binary64 (the "P64" build of the
 *     reference: the parity target); the kidmp32_* entries take binary32 arrays (KiD's default REAL).
 *   - A column profile is nz contiguous values, level kts first (k fastest),
 *     i.e. exactly KiD's `theta(k,i)` storage (W:60-93).  A batch of ncol
 *     columns is x[col*nz + k].
 *   - Every function returns 0 on success or a negative KIDMP_E* code; it
 *     never aborts and never throws.  kidmp_last_error() gives the text.
 *   - Every entry point makes the context's device current for the duration of the call and restores the
 *     caller's; device pointers must belong to the context's device (KIDMP_EINVAL otherwise).
 *   - One context per process and device; calls on one context are
 *     serialised by the caller (the reference is single-threaded, M:386-430),
 *     and its asynchronous launches must not overlap on the device (enqueue
 *     them on one stream): a context owns one work buffer.
 */
#ifndef KIDMP_H
#define KIDMP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KIDMP_OK            0
#define KIDMP_EINVAL       -1   /* bad argument (null pointer, nz out of range ...) */
#define KIDMP_ENODEV       -2   /* no HIP device / wrong architecture            */
#define KIDMP_EHIP         -3   /* a HIP runtime call failed                      */
#define KIDMP_ENOMEM       -4
#define KIDMP_ESTATE       -5   /* context not initialised / already finalised    */
#define KIDMP_EIO          -6   /* cache file missing, unreadable or malformed    */

#define KIDMP_MAX_NZ      256   /* levels per column supported by the kernels     */
#define KIDMP_NRATES       36   /* process-rate diagnostics, order of M:2967-3119 */

typedef struct kidmp_ctx kidmp_ctx;

/* Run-time switches the reference reads from KiD modules:
 *   iiwarm, set_Nc      `namelists` (M:22)
 *   l_sediment          `switches`  (M:20)  gates ice/snow/graupel fall only
 *   device              HIP device ordinal (one process per GPU)
 *   is_aerosol_aware    `module_mp_thompson09n`'s public LOGICAL (M:28)          */
typedef struct kidmp_cfg {
    int32_t iiwarm;
    int32_t l_sediment;
    double  set_Nc;        /* cloud droplet number, cm^-3 (Nt_c = set_Nc*1e6, M:381) */
    int32_t device;
    int32_t is_aerosol_aware;   /* the module's public switch (M:28; .false. in KiD): prognostic droplet number, aerosol
                                   activation / DeMott / Koop nucleation, scavenging, droplet evaporation */
} kidmp_cfg;

/* thompson_init (M:374-797): computes the gamma/rate constants on the host and
 * builds the lookup tables (M:3698-4343) with HIP kernels into HBM. */
int kidmp_init(const kidmp_cfg *cfg, kidmp_ctx **ctx_out);
void kidmp_finalize(kidmp_ctx *ctx);
const char *kidmp_last_error(const kidmp_ctx *ctx);

/* mp_thompson (M:1156-3688), host arrays, one column: the compatibility entry
 * behind the Fortran `mp_thompson` shim.  The 12 profiles are INOUT, p1d/w1d/
 * dzq IN, ppt[4] = {pptrain, pptsnow, pptgraul, pptice} INOUT (accumulated),
 * as in the reference dummy list (M:1156-1177). */
int kidmp_column_step(kidmp_ctx *ctx, int32_t nz, double dt,
                      double *qv1d, double *qc1d, double *qi1d, double *qr1d,
                      double *qs1d, double *qg1d, double *ni1d, double *nr1d,
                      double *nc1d, double *nwfa1d, double *nifa1d, double *t1d,
                      const double *p1d, const double *w1d, const double *dzq,
                      double *ppt);

/* The `do i=1,nx` loop of the KiD adapter (W:54-246) as ONE batched call on
 * host arrays x[col*nz+k]; ppt is [ncol][4] (INOUT).  rates may be NULL, else
 * receives [ncol][KIDMP_NRATES][nz] (the save_dg values of M:2962-3124).
 * Arrays KiD itself never fills may be left out (NULL) in all three host-array entries; they then neither cross
 * PCIe nor come back:
 *   nc, nwfa, nifa (all three or none)   contexts without is_aerosol_aware: the adapter passes them unset (W:36);
 *                                        the non-aerosol defaults of M:958-964 are formed on the device
 *   qi, qs, qg, ni (all four or none)    iiwarm contexts: a warm run keeps the frozen species at zero (W:46-52)
 *   w                                    contexts without is_aerosol_aware (never read)
 * A context that reads them (aerosol-aware / mixed-phase) refuses the call instead. */
int kidmp_batch_step_host(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt,
                          double *qv, double *qc, double *qi, double *qr,
                          double *qs, double *qg, double *ni, double *nr,
                          double *nc, double *nwfa, double *nifa, double *t,
                          const double *p, const double *w, const double *dz,
                          double *ppt, double *rates);

/* Same, also returning the substep counts: nstep (may be NULL) receives [ncol][4] int32 (rain, ice, snow,
 * graupel; M:3365,3447,3504,3553).  All four are 0 for a column that left through the `no_micro` early return
 * (M:1540) -- such a column makes no save_dg call in the reference, which is how the Fortran drop-in knows to
 * skip it when it replays the rate diagnostics. */
int kidmp_batch_step_host_diag(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt,
                               double *qv, double *qc, double *qi, double *qr,
                               double *qs, double *qg, double *ni, double *nr,
                               double *nc, double *nwfa, double *nifa, double *t,
                               const double *p, const double *w, const double *dz,
                               double *ppt, double *rates, int32_t *nstep);

/* Same, on DEVICE pointers (state resident in HBM), enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the null stream).  Asynchronous.
 * nstep (may be NULL) receives [ncol][4] int32 substep counts (rain, ice,
 * snow, graupel; M:3365,3447,3504,3553). */
int kidmp_batch_step_device(kidmp_ctx *ctx, int64_t ncol, int32_t nz, double dt,
                            double *qv, double *qc, double *qi, double *qr,
                            double *qs, double *qg, double *ni, double *nr,
                            double *nc, double *nwfa, double *nifa, double *t,
                            const double *p, const double *w, const double *dz,
                            double *ppt, double *rates, int32_t *nstep,
                            void *stream);

/* ---- The same three entries on binary32 arrays: the reference's `REAL` when KiD is built with its default
 * 4-byte reals (M:1168-1177 declares every dummy REAL).  `arith` selects the arithmetic inside the kernel:
 *   KIDMP_ARITH_P32N  the reference as shipped: everything it declares REAL is binary32, everything it declares
 *                     DOUBLE PRECISION (the ~70 process rates M:1184-1211, ilamr/ilamg/N0_r/N0_g M:1225, lamc/lamr/
 *                     lamg/lami/N0_exp M:1235-1236, the lookup tables) stays binary64;
 *   KIDMP_ARITH_F32   everything binary32 (the cheap end of the precision sweep of BASELINE config 5).
 * rates (may be NULL) is binary64 in both.  The double entries above are the parity build (P64). */
#define KIDMP_ARITH_P32N 0
#define KIDMP_ARITH_F32  1
int kidmp32_column_step(kidmp_ctx *ctx, int32_t nz, float dt,
                        float *qv1d, float *qc1d, float *qi1d, float *qr1d,
                        float *qs1d, float *qg1d, float *ni1d, float *nr1d,
                        float *nc1d, float *nwfa1d, float *nifa1d, float *t1d,
                        const float *p1d, const float *w1d, const float *dzq,
                        float *ppt, int32_t arith);
int kidmp32_batch_step_host(kidmp_ctx *ctx, int64_t ncol, int32_t nz, float dt,
                            float *qv, float *qc, float *qi, float *qr,
                            float *qs, float *qg, float *ni, float *nr,
                            float *nc, float *nwfa, float *nifa, float *t,
                            const float *p, const float *w, const float *dz,
                            float *ppt, double *rates, int32_t *nstep, int32_t arith);
int kidmp32_batch_step_device(kidmp_ctx *ctx, int64_t ncol, int32_t nz, float dt,
                              float *qv, float *qc, float *qi, float *qr,
                              float *qs, float *qg, float *ni, float *nr,
                              float *nc, float *nwfa, float *nifa, float *t,
                              const float *p, const float *w, const float *dz,
                              float *ppt, double *rates, int32_t *nstep,
                              int32_t arith, void *stream);

/* DEPRECATED, kept for hosts linked against earlier builds: rounds 1-2 reserved a per-batch work profile here.  The step
 * owns no per-batch device memory any more; the call checks its arguments, does nothing and returns KIDMP_OK. */
int kidmp_reserve(kidmp_ctx *ctx, int64_t ncol, int32_t nz);

/* The device entries own no per-batch device memory and never allocate: they can be captured into a hipGraph as
 * they are, and one context may have STEP launches in flight on several streams.  (The diagnostics entries further
 * down -- kidmp_reduce_rates_device, kidmp_sanity_device -- use one scratch buffer per context, allocated by
 * kidmp_init: they never allocate either, but calls of one of them on ONE context must be enqueued on one stream.) */

/* ---- host memory for the host-array entries (kidmp_batch_step_host*, kidmp32_batch_step_host) ----
 * Those entries stand where the reference's `do i=1,nx` loop works on the model's own arrays (W:54-246), so every
 * call moves the state across PCIe: 14 profiles in, 12 out per column (about 25 KB in binary64).  They run as a
 * three-stage pipeline over column chunks -- upload of chunk i+1, step of chunk i, download of chunk i-1 on three
 * streams -- which only overlaps if the DMA engines can reach the host arrays, i.e. if they are page-locked.
 * kidmp_host_alloc returns page-locked memory (NULL on failure, message in kidmp_last_error(NULL)); free it with
 * kidmp_host_free (memory the caller got from hipHostMalloc serves as well; for hipHostRegister see INTEGRATION.md 4).  Pageable
 * arrays are accepted too: results are the same, the copies are then staged by the HIP runtime and do not overlap.
 * Neither function needs a context. */
void *kidmp_host_alloc(size_t bytes);
void kidmp_host_free(void *p);
/* Columns per pipeline chunk of this context's host-array entries; 0 (default) = a quarter of the batch, rounded up
 * to 256, at most 8 192 (one chunk for batches of <= 2048 columns). */
int kidmp_set_host_chunk(kidmp_ctx *ctx, int64_t ncol_per_chunk);

/* Non-aerosol defaults for nc1d/nwfa1d/nifa1d, which the KiD wrapper leaves
 * unset (W:36): nc=Nt_c/rho, nwfa=11.1e6/rho, nifa=naIN1*0.01/rho with
 * rho=0.622p/(R T (qv+0.622)) (M:958-964).  Device pointers, [ncol*nz]. */
int kidmp_default_aerosols_device(kidmp_ctx *ctx, int64_t n,
                                  const double *qv, const double *t, const double *p,
                                  double *nc, double *nwfa, double *nifa, void *stream);

/* Domain sums of the surface precipitation (the analogue of the nx-means of
 * W:248-275): out[4] (device pointer) = sum over columns of ppt[col][0..3].
 * Multi-GPU callers all-reduce out[4] with RCCL afterwards. */
int kidmp_reduce_ppt_device(kidmp_ctx *ctx, int64_t ncol, const double *ppt,
                            double *out4, void *stream);

/* The same four sums EXACTLY, whatever the order of the additions and however the columns are spread over devices:
 * limbs (device, int64[KIDMP_PPT_LIMBS]) receives fixed-point accumulators (per species 6 limbs, limb j weighs
 * 2**(32j-128); values below 2**-128 are dropped, |x| must stay below 2**32).  Integer sums are associative, so
 * multi-GPU callers all-reduce(SUM, int64) the limbs and every partition of the columns yields the same bits;
 * kidmp_ppt_limbs_to_sums turns (host) limbs into the four doubles. */
#define KIDMP_PPT_LIMBS 24
int kidmp_reduce_ppt_exact_device(kidmp_ctx *ctx, int64_t ncol, const double *ppt, int64_t *limbs, void *stream);
int kidmp_ppt_limbs_to_sums(const int64_t *limbs, double *out4);

/* ---- several GPUs behind one call: the `do i=1,nx` loop of the KiD adapter (W:54-246) over a device list ----
 * The reference's compiled-out decomposition (M:3744-3746, M:3813-3819) splits work over MPI ranks; KiD itself is one
 * process.  kidmp_init_multi builds one context per entry of devices[] (thompson_init on every device, tables built per
 * device); kidmp_batch_step_host_multi cuts the ncol host columns into contiguous ranges (kidmp_shard_bounds: sizes
 * differ by at most one, range i on devices[i]), runs every range through its context's own upload / step / download
 * pipeline concurrently (one host thread per context), and returns in precip_sums[4] (may be NULL) the domain sums of
 * ppt -- the numerators of the nx-means of W:248-303 -- reduced over the devices with ONE RCCL all-reduce (int64 SUM of
 * the exact accumulators above; librccl.so is dlopen'ed by kidmp_init_multi, a one-GPU host never needs it).  No halo,
 * no other exchange.  Results per column are those of kidmp_batch_step_host_diag bit for bit, the sums are identical
 * for every device list.  A device may be named twice (two contexts on one card).  Optional arrays as above. */
typedef struct kidmp_multi kidmp_multi;
#define KIDMP_MAX_DEVICE_LIST 16     /* entries of devices[]; a process may hold 32 contexts at once */
int kidmp_init_multi(const kidmp_cfg *cfg, int32_t ndev, const int32_t *devices, kidmp_multi **out);   /* cfg->device is ignored */
void kidmp_finalize_multi(kidmp_multi *m);
const char *kidmp_multi_last_error(const kidmp_multi *m);
int32_t kidmp_multi_size(const kidmp_multi *m);
kidmp_ctx *kidmp_multi_context(kidmp_multi *m, int32_t i);         /* e.g. for kidmp_load_table_cache on every device */
int kidmp_shard_bounds(int64_t ncol, int32_t nshard, int32_t shard, int64_t *lo, int64_t *hi);   /* no GPU needed */
int kidmp_batch_step_host_multi(kidmp_multi *m, int64_t ncol, int32_t nz, double dt,
                                double *qv, double *qc, double *qi, double *qr,
                                double *qs, double *qg, double *ni, double *nr,
                                double *nc, double *nwfa, double *nifa, double *t,
                                const double *p, const double *w, const double *dz,
                                double *ppt, double *rates, int32_t *nstep, double *precip_sums);
/* The same call with the sanity scan of the scheme's 3-D driver (M:1025-1094) over the END state of all ncol columns:
 * sanity15 (host, may be NULL) = what kidmp_sanity_device returns for the unsharded batch -- [0..6] maxima of qc, qr, nr,
 * qs, qi, qg, ni, [7..14] numbers of negative entries of qc, qr, nr, qs, qi, qg, ni, qv --, scanned chunk by chunk on each
 * device as the chunks leave the step (exact integer atomics) and reduced over the devices beside the precipitation
 * limbs, in the same RCCL group: all-reduce(uint64, MAX) of the seven maxima (non-negative doubles order like their bit
 * patterns), all-reduce(uint64, SUM) of the eight counts.  Identical for every device list. */
int kidmp_batch_step_host_multi_diag(kidmp_multi *m, int64_t ncol, int32_t nz, double dt,
                                     double *qv, double *qc, double *qi, double *qr,
                                     double *qs, double *qg, double *ni, double *nr,
                                     double *nc, double *nwfa, double *nifa, double *t,
                                     const double *p, const double *w, const double *dz,
                                     double *ppt, double *rates, int32_t *nstep, double *precip_sums, double *sanity15);

/* Optional domain diagnostics beyond the four precipitation sums.
 * kidmp_reduce_rates_device: out[KIDMP_NRATES*nz] (device) = sum over columns of rates[col][r][k] -- the mean
 *   process-rate profiles KiD plots are these sums / ncol (save_dg(k, value, ...) of M:2967-3119, averaged over nx);
 *   fixed summation order, so the result is reproducible.  Multi-GPU callers all-reduce(SUM) it.
 * kidmp_sanity_device: the scan the scheme's own 3-D driver runs after each column (M:1025-1094):
 *   out15[0..6] = max over all n = ncol*nz entries of qc, qr, nr, qs, qi, qg, ni; out15[7..14] = how many entries of
 *   qc, qr, nr, qs, qi, qg, ni, qv are negative (the reference formats a WARNING for each).  Multi-GPU callers
 *   all-reduce(MAX) the first seven and all-reduce(SUM) the rest.
 * Both use per-context scratch from kidmp_init (see above): per context, enqueue them on one stream. */
int kidmp_reduce_rates_device(kidmp_ctx *ctx, int64_t ncol, int32_t nz, const double *rates, double *out, void *stream);
int kidmp_sanity_device(kidmp_ctx *ctx, int64_t n, const double *qc, const double *qr, const double *nr,
                        const double *qs, const double *qi, const double *qg, const double *ni, const double *qv,
                        double *out15, void *stream);

/* calc_effectRad (M:4834-4935): radiation effective radii of cloud water, cloud ice and snow, consistent with the
 * scheme's size distributions.  n = ncol*nz elements, device pointers; re_qc/re_qi/re_qs are INOUT like the reference's
 * (a level without the species keeps the caller's value; the scheme's 3-D driver presets 2.49E-6, 4.99E-6, 9.99E-6 m
 * and clamps afterwards, M:1111-1121). */
int kidmp_effective_radii_device(kidmp_ctx *ctx, int64_t n, const double *t, const double *p, const double *qv,
                                 const double *qc, const double *nc, const double *qi, const double *ni,
                                 const double *qs, double *re_qc, double *re_qi, double *re_qs, void *stream);

/* Introspection for parity tests: copy a lookup table / constant array to the
 * host.  Names are the reference's (tcg_racg ... t_Efsw; cre, crg, Dr ...).
 * Returns the number of doubles (<0 on error); out may be NULL to query. */
int64_t kidmp_get_table(kidmp_ctx *ctx, const char *name, double *out, int64_t cap);
int64_t kidmp_get_const(kidmp_ctx *ctx, const char *name, double *out, int64_t cap);

/* Interop with the reference's table cache (the only checkpoint-like facility of the path, M:3717-3829 and
 * M:3864-4078): list-directed text files run_data/racg_thompson09.data (6 tables) and
 * run_data/racs_thompson09.data (12 tables).  kidmp_save_table_cache writes the GPU-built tables so that a
 * stock KiD run with l_reuse_thompson_lookup=.true. can read them; kidmp_load_table_cache replaces this
 * context's rain-graupel / rain-snow tables by the files' contents (e.g. written by the reference).
 * `dir` is the directory holding the two files (the reference hard-codes "run_data"). */
int kidmp_save_table_cache(kidmp_ctx *ctx, const char *dir);
int kidmp_load_table_cache(kidmp_ctx *ctx, const char *dir);
/* thompson_init's own use of the two files, as the reference does it per file (M:3717-3729 / M:3822-3829 for racg,
 * M:3864-3895 / M:4065-4078 for racs): if the file exists AND l_reuse (KiD's switch l_reuse_thompson_lookup, M:20) its
 * tables replace the GPU-built ones; otherwise the GPU-built tables are written to it (write_if_built = 0 skips that,
 * e.g. on the second and later devices of a multi-GPU host).  *status (may be NULL): bit 0 / 1 = racg / racs read from
 * file, bit 2 / 3 = racg / racs written.  A missing directory is not an error here (the reference aborts, M:3718):
 * nothing is written.  No-op for an iiwarm context (M:773). */
int kidmp_table_cache_reuse(kidmp_ctx *ctx, const char *dir, int32_t l_reuse, int32_t write_if_built, int32_t *status);
/* The same format on host buffers (no GPU needed): ntab arrays of n_each doubles, Fortran element order. */
int kidmp_cache_write_file(const char *path, int32_t ntab, const double *const *tabs, int64_t n_each);
int kidmp_cache_read_file(const char *path, int32_t ntab, double *const *tabs, int64_t n_each);

/* Diagnostics: evaluate one of the column kernel's own fp64 math helpers (kid_amd/csrc/fastmath.h, which stand
 * in for the reference's DLOG / ALOG10 / EXP / 10.**x / x**y / SQRT / x**(1./3.)) on the device, elementwise on
 * host arrays: out[i] = fn(x[i]) (fn(x[i], y[i]) for KIDMP_MATH_POW; y is read but ignored otherwise). */
#define KIDMP_MATH_LOG    0
#define KIDMP_MATH_LOG10  1
#define KIDMP_MATH_EXP    2
#define KIDMP_MATH_EXP10  3
#define KIDMP_MATH_SQRT   4
#define KIDMP_MATH_CBRT   5
#define KIDMP_MATH_POW    6
#define KIDMP_MATH_RCP_SEED 7  /* raw v_rcp_f64(y)                                  */
#define KIDMP_MATH_DIV    8    /* the kernel's x / y (seed, one Newton step, residual correction) */
#define KIDMP_MATH_IEEE_DIV 9  /* IEEE x / y, for comparison                        */
#define KIDMP_MATH_RCP    10   /* the kernel's 1 / y (seed, one cubically convergent step) */
int kidmp_math_probe(kidmp_ctx *ctx, int32_t fn, int64_t n, const double *x, const double *y, double *out);

/* Seconds spent in table construction during kidmp_init (host wall clock). */
double kidmp_init_seconds(const kidmp_ctx *ctx);

/* Name of the column-step kernel as it appears in rocprofv3 traces. */
const char *kidmp_kernel_name(void);

/* Identifies the code object of this context's nz <= 120 column-step kernel:
 * "src:<hash of the kernel sources and flags>;vgpr:<n>;lds:<bytes>;scratch:<bytes>".  Profiles under profiles/
 * carry it, so a counter file measured on another build of the kernel is recognised as stale (bench.py). */
const char *kidmp_kernel_fingerprint(kidmp_ctx *ctx);
/* The same for the binary32 code objects behind kidmp32_*: arith = KIDMP_ARITH_P32N or KIDMP_ARITH_F32 ("" otherwise). */
const char *kidmp32_kernel_fingerprint(kidmp_ctx *ctx, int32_t arith);

#ifdef __cplusplus
}
#endif
#endif /* KIDMP_H */
