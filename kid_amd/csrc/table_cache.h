// table_cache.h -- read/write the reference's run_data/*.data list-directed table caches (M:3717-3829, M:3864-4078).
#pragma once
#include <cstdint>

namespace kidmp {
int cache_write(const char *path, int ntab, const double *const *tabs, int64_t n_each);
int cache_read(const char *path, int ntab, double *const *tabs, int64_t n_each);
}
