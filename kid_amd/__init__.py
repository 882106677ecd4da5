"""kid_amd -- MI355X-native Thompson-09n column microphysics for the KiD driver.

One hot path of EnverRamirez/KiD (module_mp_thompson09n.f90 / mphys_thompson09n.f90)
as hand-written HIP kernels for gfx950 behind a C ABI (include/kidmp.h).
"""
from .thompson import (KidmpError, ThompsonMP, ThompsonMulti, mp_thompson, thompson_init, STATE_NAMES,  # noqa: F401
                       FORCING_NAMES, RATE_NAMES, lib_path, load_library, cache_read_file, cache_write_file,
                       limbs_to_sums, shard_bounds)
