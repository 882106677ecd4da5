"""The HIP path against the reference's OWN recorded outputs (-m gpu): the known-answer soundings of SURVEY.md section 9h
(KAT-A warm / mixed over 200 calls, KAT-C over 3 calls) through kidmp_column_step, every quoted digit (6-7).  These are
the only reference-generated values there are (the mount holds no fixtures and the reference cannot be rebuilt here):
tests/test_oracle_kat.py pins the oracle on them, this file pins the product path on them directly.  KAT-B (the KiD
adapter, 360 steps) runs through the Fortran drop-in in tests/test_fortran_gpu.py."""
import numpy as np
import pytest

import kat_cases as kc
from kid_amd import STATE_NAMES

pytestmark = pytest.mark.gpu
ARGS = STATE_NAMES + ("p", "w", "dz")


def _close(x, ref, digits):
    return abs(x / ref - 1.0) < 1.0 * 10.0 ** (1 - digits)     # one unit in the last quoted place


def _run(m, st, nsteps):
    hist = []
    for _ in range(nsteps):
        ppt = m.mp_thompson(*[st[k] for k in ARGS], dt=10.0)    # precipitation arguments zeroed before each call
        hist.append(np.array(ppt))
    return hist


def test_kat_a_warm_200_calls_match_the_reference_digits(gpu_warm):
    st = kc.kat_a(False)
    hist = _run(gpu_warm, st, 200)
    assert _close(st["qv"].sum(), 4.88044e-1, 6)
    assert _close(st["qc"].sum(), 6.70530e-3, 6)
    assert _close(st["qr"].sum(), 7.32289e-4, 6)
    assert st["qi"].sum() == 0 and st["qs"].sum() == 0 and st["qg"].sum() == 0
    assert _close(hist[-1][0], 3.254535e-4, 7)
    assert _close(st["qr"][0], 9.58594e-6, 6)
    assert abs(st["t"][0] - 297.706) < 1e-3
    assert all(h[0] == 0.0 for h in hist[:3]) and all(np.all(h[1:] == 0.0) for h in hist)


def test_kat_a_mixed_200_calls_match_the_reference_digits(gpu_mixed):
    st = kc.kat_a(True)
    hist = _run(gpu_mixed, st, 200)
    for k, v in dict(qv=5.13751e-1, qc=8.00777e-4, qr=9.15208e-3, qi=3.76939e-4, qs=4.26247e-2, qg=3.18060e-3).items():
        assert _close(st[k].sum(), v, 6), (k, st[k].sum())
    assert _close(hist[-1][0], 1.708898e-2, 7)
    assert _close(st["qr"][0], 2.84659e-4, 6)
    assert abs(st["t"][0] - 295.999) < 1e-3
    for i in (0, 1, 2, 199):
        assert np.all(hist[i][1:] == 0.0)                      # pptsnow / graul / ice are 0 at calls 1-3 and 200
    assert all(h[0] == 0.0 for h in hist[:3])


def test_kat_c_three_calls_match_the_reference_digits(gpu_mixed):
    st = kc.kat_c()
    hist = _run(gpu_mixed, st, 3)
    for ppt, (pr, pg) in zip(hist, [(5.97724e-1, 4.64617e-1), (7.93538e-1, 2.34151e-1), (9.32341e-1, 6.68447e-2)]):
        assert _close(ppt[0], pr, 6) and _close(ppt[2], pg, 6), ppt
