"""The C-ABI library: it loads, exports every symbol include/kidmp.h declares, and refuses
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "kidmp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(kidmp_[a-z_0-9]+)\s*\(", hdr)))


def test_header_declares_the_expected_entry_points():
    syms = _declared_symbols()
    for s in ("kidmp_init", "kidmp_finalize", "kidmp_column_step", "kidmp_batch_step_host",
              "kidmp_batch_step_device", "kidmp_default_aerosols_device", "kidmp_reduce_ppt_device",
              "kidmp_get_table", "kidmp_get_const", "kidmp_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from kid_amd import lib_path
    lib = ctypes.CDLL(lib_path())
    for s in _declared_symbols():
        assert hasattr(lib, s), s


def test_no_signature_uses_torch_or_cxx_types():
    hdr = open(os.path.join(ROOT, "include", "kidmp.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only, comments stripped
    assert "torch" not in code and "std::" not in code and "hipStream_t" not in code and "#include <hip" not in code


def test_header_compiles_as_strict_c99_and_cxx(tmp_path):
    """include/kidmp.h is the boundary a C, C++ or Fortran (ISO_C_BINDING) host binds: it must stand alone."""
    import subprocess
    src = tmp_path / "use_kidmp.c"
    src.write_text('#include "kidmp.h"\n'
                   'int use(void) { kidmp_cfg c; c.iiwarm = 1; c.is_aerosol_aware = 0; return (int)sizeof(c) + KIDMP_NRATES + KIDMP_ARITH_F32; }\n')
    inc = os.path.join(ROOT, "include")
    for cmd in (["gcc", "-std=c99", "-pedantic"], ["g++", "-std=c++11", "-pedantic", "-x", "c++"]):
        r = subprocess.run(cmd + ["-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "o.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from kid_amd import KidmpError, ThompsonMP
    with pytest.raises(KidmpError) as e:
        ThompsonMP(iiwarm=True)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_product_never_imports_the_oracle():
    """kid_amd/ must not reference oracle/ in any form (the oracle is test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kid_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".f90", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "th_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_fortran_shim_binds_the_c_abi():
    src = open(os.path.join(ROOT, "kid_amd", "fortran", "module_mp_thompson09n.f90")).read()
    for s in ("kidmp_init", "kidmp_finalize", "kidmp_batch_step_host_diag", "kidmp_last_error"):
        assert "name='%s'" % s in src
    # the reference's dummy list, M:1156-1162
    assert re.search(r"subroutine mp_thompson \(qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, &\s*"
                     r"nr1d, nc1d, nwfa1d, nifa1d, t1d, p1d, w1d, dzq, &\s*"
                     r"pptrain, pptsnow, pptgraul, pptice, &\s*kts, kte, dt, ii, jj\)", src)


# ---- host-side pieces of the multi-GPU entry (no GPU needed) ----
def test_shard_bounds_partition_the_columns():
    """kidmp_shard_bounds: contiguous, exhaustive, sizes differ by at most one, identical to kid_amd.sharding's."""
    from kid_amd import sharding, thompson
    for ncol in (0, 1, 7, 8, 9, 1000, 10 ** 6, 10 ** 6 + 5):
        for n in (1, 2, 3, 8):
            b = [thompson.shard_bounds(ncol, n, i) for i in range(n)]
            assert b[0][0] == 0 and b[-1][1] == ncol
            assert all(b[i][1] == b[i + 1][0] for i in range(n - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
            assert b == [sharding.shard_bounds(ncol, i, n) for i in range(n)]
    with pytest.raises(thompson.KidmpError):
        thompson.shard_bounds(10, 2, 2)


def test_exact_precipitation_limbs_convert_like_exact_rationals():
    """kidmp_ppt_limbs_to_sums: limb j of a species weighs 2**(32 j - 128); un-normalised limbs (what an all-reduce
    of many partial sums leaves: each up to 2**63) must give the same doubles as the exact rational sum."""
    import random
    from fractions import Fraction
    import numpy as np
    from kid_amd import thompson
    rnd = random.Random(7)
    for trial in range(200):
        limbs, want = [], []
        for sp in range(4):
            total = 0
            mine = []
            for j in range(6):
                v = rnd.randrange(-2 ** 62, 2 ** 62) if trial % 2 else rnd.randrange(0, 2 ** 32)
                if j >= 4 and trial % 3 == 0:
                    v = 0
                mine.append(v)
                total += v << (32 * j)
            limbs += mine
            want.append(float(Fraction(total, 2 ** 128)))
        got = thompson.limbs_to_sums(np.array(limbs, dtype=np.int64))
        for g, w in zip(got, want):
            assert g == w or abs(g - w) <= abs(w) * 2.3e-16, (g, w)     # long-double summation: within one rounding
    # and two different splittings of one total agree bit for bit
    a = np.zeros(24, dtype=np.int64); b = np.zeros(24, dtype=np.int64)
    a[0], a[1] = 2 ** 40 + 5, 7
    b[0], b[1] = 5, 7 + 2 ** 8                                       # 2**40 = 2**8 * 2**32 carried into limb 1
    assert np.array_equal(thompson.limbs_to_sums(a), thompson.limbs_to_sums(b))
