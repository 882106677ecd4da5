// host_asan_check.cpp -- the host-side C++ of the product under AddressSanitizer + UBSan (make -C kid_amd/csrc asan).
// No GPU: the host half of thompson_init (thompson_host_init.cpp) for several switch settings, and the reader/writer
// of the reference's table-cache format (table_cache.cpp) on well-formed, Fortran-spelled, truncated and garbage files.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "table_cache.h"
#include "thompson_host_init.h"

using namespace kidmp;

static int fails = 0;
#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); ++fails; } } while (0)

static void write_text(const char *path, const std::string &s)
{
    FILE *f = std::fopen(path, "w");
    if (!f) { std::perror(path); std::exit(2); }
    std::fwrite(s.data(), 1, s.size(), f);
    std::fclose(f);
}

int main()
{
    // ---- thompson_init's host half, M:374-670, for every switch combination and a few droplet numbers ----
    for (int iiwarm = 0; iiwarm < 2; ++iiwarm)
        for (int sed = 0; sed < 2; ++sed)
            for (double nc : {10.0, 100.0, 750.0, 3000.0}) {
                static Consts c;
                static Bins b;
                std::memset(&c, 0xff, sizeof c);
                std::memset(&b, 0xff, sizeof b);
                host_init(iiwarm, sed, nc, c, b);
                CHECK(c.iiwarm == iiwarm && c.l_sediment == sed);
                CHECK(c.Nt_c == nc * 1.e6);
                for (int i = 0; i < 13; ++i) CHECK(std::isfinite(c.cre[i]) && std::isfinite(c.crg[i]) && c.crg[i] > 0.);
                for (int i = 0; i < 18; ++i) CHECK(std::isfinite(c.cse[i]) && std::isfinite(c.csg[i]));
                for (int i = 0; i < nbins; ++i) CHECK(b.Dr[i] > 0. && b.Ds[i] > 0. && b.Dg[i] > 0. && b.Di[i] > 0. && b.Dc[i] > 0.);
                for (int i = 1; i < nbins; ++i) CHECK(b.Dr[i] > b.Dr[i - 1]);
                CHECK(b.r_c[0] == 1.e-6 && b.r_r[ntb_r - 1] == 1.e-2);
            }

    // ---- table cache: round trip, Fortran spellings, short and malformed input ----
    const int64_t n = 1000;
    std::vector<double> a(n), b2(n), ra(n), rb(n);
    for (int64_t i = 0; i < n; ++i) { a[i] = std::ldexp(1.0 + i * 1e-3, int(i % 600) - 300); b2[i] = -a[i] / 3.0; }
    a[7] = 0.0; a[8] = 5e-324; a[9] = 1.7976931348623157e308;
    const double *w[2] = {a.data(), b2.data()};
    double *r[2] = {ra.data(), rb.data()};
    CHECK(cache_write("rt.data", 2, w, n) == 0);
    CHECK(cache_read("rt.data", 2, r, n) == 0);
    CHECK(std::memcmp(a.data(), ra.data(), n * sizeof(double)) == 0 && std::memcmp(b2.data(), rb.data(), n * sizeof(double)) == 0);
    CHECK(cache_read("does_not_exist.data", 2, r, n) == -1);
    CHECK(cache_write("no_such_dir/x.data", 2, w, n) == -1);
    // asking for more values than the file holds
    CHECK(cache_read("rt.data", 3, r, n) != 0 || true);                       // (r has two tables: only the count matters below)
    {
        std::vector<double> big(3 * n);
        double *r3[3] = {big.data(), big.data() + n, big.data() + 2 * n};
        CHECK(cache_read("rt.data", 3, r3, n) == -3);
    }
    // what a Fortran processor may write: repeats, D exponents, omitted exponent letter, commas, odd line breaks
    write_text("f.data", "  3*1.5D0, 2*,\n 1.0-310 2.5E+01\n\n,7.0d-2   4*0.125\n");
    {
        double v[12];
        double *rr[1] = {v};
        CHECK(cache_read("f.data", 1, rr, 12) == 0);
        const double want[12] = {1.5, 1.5, 1.5, 0., 0., 1.0e-310, 25.0, 0.07, 0.125, 0.125, 0.125, 0.125};
        for (int i = 0; i < 12; ++i) CHECK(v[i] == want[i]);
        CHECK(cache_read("f.data", 1, rr, 13) == -3);                          // one value short
    }
    for (const char *junk : {"", "   \n\n", "abc", "1.0 2.0 x3", "-3*1.0", "0*1.0", "1e", "*", "1.0,,,,", "99999999999999999999*1"}) {
        write_text("junk.data", junk);
        double v[4] = {0, 0, 0, 0};
        double *rr[1] = {v};
        const int rc = cache_read("junk.data", 1, rr, 4);
        CHECK(rc == -3 || rc == 0);                                            // never a crash, never out of bounds
    }
    // a long token and a file without a trailing newline
    write_text("long.data", std::string(100000, '1') + " 2");
    {
        double v[2];
        double *rr[1] = {v};
        CHECK(cache_read("long.data", 1, rr, 2) == 0 && v[1] == 2.0 && std::isinf(v[0]));
    }
    if (fails) { std::fprintf(stderr, "host_asan_check: %d check(s) failed\n", fails); return 1; }
    std::puts("host_asan_check ok");
    return 0;
}
