// table_cache.cpp -- interop with the reference's lookup-table cache files.
//
// The reference saves the two expensive table families as Fortran list-directed text:
//     run_data/racg_thompson09.data   write(12,*) tcg_racg, tmr_racg, tcr_gacr, tmg_gacr, tnr_racg, tnr_gacr   (M:3823-3828)
//     run_data/racs_thompson09.data   write(13,*) tcs_racs1, tmr_racs1, tcs_racs2, tmr_racs2, tcr_sacr1, tms_sacr1,
//                                                 tcr_sacr2, tms_sacr2, tnr_racs1, tnr_racs2, tnr_sacr1, tnr_sacr2 (M:4066-4077)
// and reads them back with read(12,*) / read(13,*) when l_reuse_thompson_lookup is set (M:3721-3727, M:3883-3894).
// One `write(u,*) array` statement = all elements in array-element (column-major) order, separated by blanks
// and/or commas, wrapped over records; list-directed input additionally accepts `r*c` repeat forms, `D`
// exponents and arbitrary line breaks.  This file writes that format (17 significant digits, so binary64
// round-trips exactly) and parses everything a Fortran processor may have written.
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "table_cache.h"

namespace kidmp {

int cache_write(const char *path, int ntab, const double *const *tabs, int64_t n_each)
{
    FILE *f = std::fopen(path, "w");
    if (!f) return -1;
    for (int t = 0; t < ntab; ++t) {
        for (int64_t i = 0; i < n_each; ++i) {
            // three values per record, like a list-directed REAL(8) record of ~80 columns
            if (std::fprintf(f, " %.17g%s", tabs[t][i], (i % 3 == 2 || i + 1 == n_each) ? "\n" : "") < 0) {
                std::fclose(f);
                return -2;
            }
        }
    }
    return std::fclose(f) == 0 ? 0 : -2;
}

// Reads ntab*n_each values in order.  Returns 0, -1 (cannot open), -3 (malformed / too few values).
int cache_read(const char *path, int ntab, double *const *tabs, int64_t n_each)
{
    FILE *f = std::fopen(path, "r");
    if (!f) return -1;
    std::string buf;
    {
        char chunk[1 << 16];
        size_t n;
        while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, n);
    }
    std::fclose(f);
    const int64_t total = int64_t(ntab) * n_each;
    int64_t got = 0;
    const char *p = buf.c_str(), *end = p + buf.size();
    auto put = [&](double v) {
        if (got < total) tabs[got / n_each][got % n_each] = v;
        ++got;
    };
    std::string tok;
    while (p < end && got < total) {
        while (p < end && (std::isspace((unsigned char)*p) || *p == ',')) ++p;       // value separators
        if (p >= end) break;
        const char *q = p;
        while (q < end && !std::isspace((unsigned char)*q) && *q != ',') ++q;
        tok.assign(p, q);
        p = q;
        long rep = 1;
        size_t star = tok.find('*');
        std::string val = tok;
        if (star != std::string::npos) {                                              // r*c repeat form
            rep = std::strtol(tok.substr(0, star).c_str(), nullptr, 10);
            val = tok.substr(star + 1);
            if (rep <= 0) return -3;
        }
        for (char &ch : val)
            if (ch == 'D' || ch == 'd' || ch == 'Q' || ch == 'q') ch = 'E';           // Fortran exponent letters
        // "1.5-310" (exponent letter omitted for 3-digit exponents) -> insert 'E'
        for (size_t i = 1; i < val.size(); ++i)
            if ((val[i] == '+' || val[i] == '-') && (std::isdigit((unsigned char)val[i - 1]) || val[i - 1] == '.')) {
                val.insert(i, "E");
                break;
            }
        char *e = nullptr;
        errno = 0;
        const double v = val.empty() ? 0.0 : std::strtod(val.c_str(), &e);            // "r*" = r nulls
        if (!val.empty() && (e == val.c_str() || *e != '\0')) return -3;
        if (rep > total - got) rep = long(total - got);                               // a huge repeat count must not spin
        for (long r = 0; r < rep; ++r) put(v);
    }
    return got >= total ? 0 : -3;
}

}  // namespace kidmp
