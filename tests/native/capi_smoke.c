/* The C ABI from plain C (c99): what INTEGRATION.md promises a C host.  Builds a few warm-rain columns, steps them
 * through kidmp_batch_step_host with the arrays KiD never fills left out, and writes its inputs (capi_in.bin: qv qc qr
 * nr t p dz) and results (capi_out.bin: qv qc qr nr t, then ppt) as raw doubles.  tests/test_gpu_capi_c.py steps the
 * same inputs through the Python wrapper and compares bit for bit.  With a device list as third argument the columns are
 * then stepped again through the library's multi-device entry and compared with the single-device run. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "kidmp.h"

int main(int argc, char **argv)
{
    const int nz = 120;
    const long ncol = argc > 1 ? atol(argv[1]) : 8;
    const int nsteps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t n = (size_t)ncol * nz;
    double *buf = (double *)kidmp_host_alloc(7 * n * sizeof(double));
    double *ppt = (double *)kidmp_host_alloc(4 * (size_t)ncol * sizeof(double));
    double *qv, *qc, *qr, *nr, *t, *p, *dz;
    kidmp_cfg cfg;
    kidmp_ctx *ctx = NULL;
    long c;
    int k, s, rc;
    if (!buf || !ppt) { fprintf(stderr, "kidmp_host_alloc: %s\n", kidmp_last_error(NULL)); return 2; }
    qv = buf; qc = buf + n; qr = buf + 2 * n; nr = buf + 3 * n; t = buf + 4 * n; p = buf + 5 * n; dz = buf + 6 * n;
    for (c = 0; c < ncol; ++c)
        for (k = 0; k < nz; ++k) {
            const double z = (k + 0.5) * 25.0, scale = 1.0 + 0.05 * (double)c;
            const size_t i = (size_t)c * nz + k;
            const int cloud = z > 800.0 && z < 2000.0;
            p[i] = 1.0e5 * pow(1.0 - 2.2557e-5 * z, 5.2559);
            t[i] = 297.0 - 6.5e-3 * z;
            qv[i] = 0.015 - 0.004 * z / 3000.0;
            qc[i] = cloud ? 8.0e-4 : 0.0;
            qr[i] = cloud ? 3.0e-4 * scale : 0.0;
            nr[i] = cloud ? 2.0e4 : 0.0;
            dz[i] = 25.0;
        }
    for (c = 0; c < 4 * ncol; ++c) ppt[c] = 0.0;
    {
        FILE *f = fopen("capi_in.bin", "wb");
        if (!f || fwrite(buf, sizeof(double), 7 * n, f) != 7 * n) return 6;
        fclose(f);
    }
    cfg.iiwarm = 1; cfg.l_sediment = 1; cfg.set_Nc = 100.0; cfg.device = 0; cfg.is_aerosol_aware = 0;
    rc = kidmp_init(&cfg, &ctx);
    if (rc) { fprintf(stderr, "kidmp_init: %d %s\n", rc, kidmp_last_error(NULL)); return 3; }
    for (s = 0; s < nsteps; ++s) {
        rc = kidmp_batch_step_host(ctx, ncol, nz, 10.0, qv, qc, NULL, qr, NULL, NULL, NULL, nr, NULL, NULL, NULL, t,
                                   p, NULL, dz, ppt, NULL);
        if (rc) { fprintf(stderr, "kidmp_batch_step_host: %d %s\n", rc, kidmp_last_error(ctx)); return 4; }
    }
    {
        double sqv = 0., sqc = 0., sqr = 0., snr = 0., st = 0., sp = 0.;
        size_t i;
        for (i = 0; i < n; ++i) { sqv += qv[i]; sqc += qc[i]; sqr += qr[i]; snr += nr[i]; st += t[i]; }
        for (c = 0; c < ncol; ++c) sp += ppt[4 * c];
        printf("CAPI %.17g %.17g %.17g %.17g %.17g %.17g\n", sqv, sqc, sqr, snr, st, sp);
    }
    {
        FILE *f = fopen("capi_out.bin", "wb");
        if (!f || fwrite(buf, sizeof(double), 5 * n, f) != 5 * n || fwrite(ppt, sizeof(double), 4 * (size_t)ncol, f) != 4 * (size_t)ncol) return 7;
        fclose(f);
    }
    /* the same columns once more from their initial state, spread over a device list by the library itself
     * (kidmp_init_multi / kidmp_batch_step_host_multi: what a single-process C or Fortran host uses for several GPUs).
     * argv[3] = the list, e.g. "0,0" (two contexts on one card) or "0,1,2,3"; results must equal the run above. */
    if (argc > 3) {
        int32_t devs[8];
        int32_t ndev = 0;
        const char *q = argv[3];
        double *buf2 = (double *)kidmp_host_alloc(7 * n * sizeof(double));
        double *ppt2 = (double *)kidmp_host_alloc(4 * (size_t)ncol * sizeof(double));
        double sums[4] = {-1., -1., -1., -1.}, direct = 0.;
        kidmp_multi *m = NULL;
        size_t i, differ = 0;
        FILE *f = fopen("capi_in.bin", "rb");
        if (!buf2 || !ppt2 || !f || fread(buf2, sizeof(double), 7 * n, f) != 7 * n) return 8;
        fclose(f);
        while (*q && ndev < 8) { devs[ndev++] = (int32_t)strtol(q, (char **)&q, 10); if (*q == ',') ++q; }
        for (c = 0; c < 4 * ncol; ++c) ppt2[c] = 0.0;
        rc = kidmp_init_multi(&cfg, ndev, devs, &m);
        if (rc) { fprintf(stderr, "kidmp_init_multi: %d %s\n", rc, kidmp_last_error(NULL)); return 9; }
        for (s = 0; s < nsteps; ++s) {
            rc = kidmp_batch_step_host_multi(m, ncol, nz, 10.0, buf2, buf2 + n, NULL, buf2 + 2 * n, NULL, NULL, NULL, buf2 + 3 * n,
                                             NULL, NULL, NULL, buf2 + 4 * n, buf2 + 5 * n, NULL, buf2 + 6 * n, ppt2, NULL, NULL, sums);
            if (rc) { fprintf(stderr, "kidmp_batch_step_host_multi: %d %s\n", rc, kidmp_multi_last_error(m)); return 10; }
        }
        for (i = 0; i < 5 * n; ++i) differ += buf2[i] != buf[i];
        for (i = 0; i < 4 * (size_t)ncol; ++i) differ += ppt2[i] != ppt[i];
        for (c = 0; c < ncol; ++c) direct += ppt2[4 * c];
        printf("MULTI %d %lu %.17g %.17g %.17g %.17g %.17g\n", (int)kidmp_multi_size(m), (unsigned long)differ, sums[0], sums[1],
               sums[2], sums[3], direct);
        kidmp_finalize_multi(m);
        kidmp_host_free(buf2);
        kidmp_host_free(ppt2);
    }
    /* a mixed-phase context refuses the lean call instead of reading address 0 */
    kidmp_finalize(ctx);
    cfg.iiwarm = 0;
    rc = kidmp_init(&cfg, &ctx);
    if (rc) return 5;
    rc = kidmp_batch_step_host(ctx, ncol, nz, 10.0, qv, qc, NULL, qr, NULL, NULL, NULL, nr, NULL, NULL, NULL, t, p, NULL, dz, ppt, NULL);
    printf("REFUSED %d %s\n", rc, kidmp_last_error(ctx));
    kidmp_finalize(ctx);
    kidmp_host_free(buf);
    kidmp_host_free(ppt);
    return 0;
}
