"""Pins the CPU oracle against the reference's own outputs.

The numbers below are what the reference Fortran (P64 build) produced for these exact
soundings during the survey (KAT-A/B/C; 6-7 significant digits).
They are the only reference-generated values available: the mount holds no tests or
golden vectors, and the reference cannot be rebuilt in this image."""
import numpy as np
import pytest

import kat_cases as kc


def _run(o, st, nsteps):
    ppt = None
    hist = []
    for _ in range(nsteps):
        ppt, _, ns, _ = o.column_step(st, 10.0)
        hist.append((ppt.copy(), ns))
    return ppt, hist


def _close(x, ref, digits):
    # the survey quotes `digits` significant digits: allow one unit in the last quoted place
    return abs(x / ref - 1.0) < 1.0 * 10.0 ** (1 - digits)


def test_kat_a_warm_200_steps(oracle_warm):
    st = kc.kat_a(False)
    ppt, hist = _run(oracle_warm, st, 200)
    assert _close(st["qv"].sum(), 4.88044e-1, 6)
    assert _close(st["qc"].sum(), 6.70530e-3, 6)
    assert _close(st["qr"].sum(), 7.32289e-4, 6)
    assert st["qi"].sum() == 0 and st["qs"].sum() == 0 and st["qg"].sum() == 0
    assert _close(ppt[0], 3.254535e-4, 7)
    assert _close(st["qr"][0], 9.58594e-6, 6)
    assert abs(st["t"][0] - 297.706) < 1e-3
    assert all(h[0][0] == 0.0 for h in hist[:3])          # pptrain is 0 at calls 1-3
    assert all(np.all(h[0][1:] == 0.0) for h in hist)     # no frozen precipitation


@pytest.mark.slow
def test_kat_a_mixed_200_steps(oracle_mixed):
    st = kc.kat_a(True)
    ppt, hist = _run(oracle_mixed, st, 200)
    for k, v in dict(qv=5.13751e-1, qc=8.00777e-4, qr=9.15208e-3, qi=3.76939e-4, qs=4.26247e-2,
                     qg=3.18060e-3).items():
        assert _close(st[k].sum(), v, 6), k
    assert _close(ppt[0], 1.708898e-2, 7)
    assert _close(st["qr"][0], 2.84659e-4, 6)
    assert abs(st["t"][0] - 295.999) < 1e-3
    for i in (0, 1, 2, 199):
        assert np.all(hist[i][0][1:] == 0.0)              # pptsnow/graul/ice are 0 at calls 1-3 and 200
    assert all(h[0][0] == 0.0 for h in hist[:3])


@pytest.mark.slow
def test_kat_c_substeps_and_precip(oracle_mixed):
    st = kc.kat_c()
    _, hist = _run(oracle_mixed, st, 3)
    exp = [(5.97724e-1, 4.64617e-1, 25), (7.93538e-1, 2.34151e-1, 24), (9.32341e-1, 6.68447e-2, 23)]
    for (ppt, ns), (pr, pg, n) in zip(hist, exp):
        assert _close(ppt[0], pr, 6) and _close(ppt[2], pg, 6)
        assert ns == [n, 1, 1, n]                         # rain, ice, snow, graupel


def test_kat_b_kid_adapter_360_steps(oracle_warm):
    c = kc.kat_b()
    nz, nx, dt = c["nz"], c["nx"], c["dt"]
    theta, qv, hy = c["theta"].copy(), c["qv"].copy(), c["hydro"].copy()
    z0, zh = np.zeros(nz * nx), np.zeros(hy.size)
    for _ in range(360):
        dth, dqv, dhy, _ = oracle_warm.kid_interface(nz, nx, dt, c["p0"], c["r_on_cp"], theta, z0, z0,
                                                     c["exner"], c["dz"], qv, z0, z0, hy, zh, zh)
        theta += dt * dth
        qv += dt * dqv
        hy += dt * dhy.reshape(hy.shape)
    assert _close(qv.sum(), 1.530434, 7)
    assert _close(hy[0, 0].sum(), 2.218719e-2, 7)
    assert _close(hy[0, 1].sum(), 2.694135e-3, 7)
    assert _close(hy[1, 1].sum(), 1.060568e6, 7)
