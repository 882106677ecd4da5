"""The oracle's P32n mode (the reference's NATIVE arithmetic: REAL = binary32 state and work variables, DOUBLE
PRECISION = binary64 process rates) pinned on the native known answers the survey recorded from the reference as
shipped: KAT-B through the adapter (360 warm steps) and KAT-A mixed (200 steps).

The native build differs from the P64 build by 7e-5 ... 3.4e-4 on these sums.  Round 4 found what had kept the P32n
oracle 1.3e-5 from the native KAT-B digits: not its arithmetic but the TEST INPUTS -- the survey's native probe forms
z, p, exner, T, theta and qv in binary32 (a default-REAL driver, powf from the same glibc), the tests had formed them
in binary64 and rounded.  With the inputs formed as the driver forms them (kat_cases.kat_b_native) the P32n oracle
reproduces the recorded native digits of KAT-B: all four sums within one unit of the seventh digit (1.3e-7 ... 3.3e-7;
a REAL SUM of 120 terms is itself uncertain by that much in its seventh digit, and the survey's sums were such).  KAT-A
mixed (tables built with P64 constants here, chaotic M:3596 levels on the way) lands within 2.1e-5 (sum qi) and 7e-7
(rain at call 200)."""
import os

import numpy as np
import pytest

import kat_cases as kc

f32 = np.float32
VIEW_NAMES = [("Nt_c", 0), ("Sc3", 0), ("D0i", 0), ("xm0s", 0), ("xm0g", 0), ("oig1", 0), ("oig2", 0), ("org1", 0),
              ("org2", 0), ("org3", 0), ("oams", 0), ("ocms", 0), ("ocmg", 0), ("ogg1", 0), ("ogg2", 0), ("ogg3", 0),
              ("t1_qr_qc", 0), ("t2_qr_qi", 0), ("t1_qg_qc", 0), ("t1_qr_ev", 0), ("t2_qr_ev", 0), ("t2_qs_sd", 0),
              ("t1_qs_me", 0), ("t2_qs_me", 0), ("t1_qg_sd", 0), ("t2_qg_sd", 0), ("t1_qg_me", 0), ("t2_qg_me", 0)] \
    + [("cie", i) for i in range(1, 8)] + [("cig", i) for i in range(1, 8)] + [("cre", i) for i in range(1, 14)] \
    + [("crg", i) for i in range(1, 14)] + [("cse", i) for i in range(1, 19)] + [("csg", i) for i in range(1, 19)] \
    + [("cge", i) for i in range(1, 13)] + [("cgg", i) for i in range(1, 13)] + [("ocg1", i) for i in range(1, 16)] \
    + [("ccg2", i) for i in range(1, 16)]


def test_constants_p64_view_is_the_context_and_p32n_view_is_binary32(oracle_warm):
    for name, i in VIEW_NAMES:
        v64 = oracle_warm.view_const(name, i)
        v32 = oracle_warm.view_const(name, i, p32n=True)
        assert v64 > -1e29 and v32 > -1e29, name
        try:                                                                     # P64: computed twice, same bits
            ctx = oracle_warm.const(name)
            assert v64 == ctx[i - 1 if len(ctx) > 1 else 0], (name, i)
        except KeyError:
            pass
        assert float(f32(v32)) == v32, (name, i)                                 # a binary32 value
        assert abs(v32 / v64 - 1.0) < 5e-6, (name, i, v32, v64)                  # EXP(GAMMLN) in REAL: 1e-7 ... 2e-6 (Gamma(19))


def test_kat_b_native_360_steps_through_the_adapter(oracle_warm):
    c = kc.kat_b_native()
    nz, nx, dt = c["nz"], 1, c["dt"]
    theta, qv, hy = c["theta"].copy(), c["qv"].copy(), c["hydro"].copy()
    exner, dz = c["exner"], c["dz"]
    z0, zh = np.zeros(nz, dtype=f32), np.zeros(hy.size, dtype=f32)
    for _ in range(360):
        dth, dqv, dhy, _ = oracle_warm.kid_interface_p32n(nz, nx, dt, c["p0"], c["r_on_cp"], theta, z0, z0, exner, dz,
                                                          qv, z0, z0, hy, zh, zh)
        theta = (theta + f32(dt) * dth).astype(f32)
        qv = (qv + f32(dt) * dqv).astype(f32)
        hy = (hy + f32(dt) * dhy.reshape(hy.shape)).astype(f32)
    got = [float(a.astype(np.float64).sum()) for a in (qv, hy[0, 0], hy[0, 1], hy[1, 1])]
    native = [1.530434, 2.218541e-2, 2.693803e-3, 1.060634e6]       # SURVEY 9h, reference as shipped (7 digits)
    p64 = [1.530434, 2.218719e-2, 2.694135e-3, 1.060568e6]          # SURVEY 9h, reference P64 build
    print("KAT-B native, P32n oracle:", ["%.7e" % g for g in got])
    for g, n in zip(got, native):
        assert abs(g / n - 1) < 5e-7, (got, native)                 # one unit of the seventh digit (measured 1.3e-7 ... 3.3e-7)
    for g, n, p in zip(got[1:], native[1:], p64[1:]):
        assert abs(g - n) < 0.02 * abs(p - n), (g, n, p)            # fifty times closer to native than P64 is


def test_native_inputs_are_the_committed_ones():
    """kat_b_native / kat_a_native call this host's powf / expf: the binary32 profiles they form are frozen in
    tests/golden/kat_native_inputs.npz (written on the image the native digits were matched on, glibc 2.35), so a libm
    whose binary32 functions round differently shows up here and not as a shifted seventh digit above."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_native_inputs.npz"))
    b, a = kc.kat_b_native(), kc.kat_a_native(True)
    for k in ("theta", "exner", "qv"):
        assert np.array_equal(b[k], g["b_" + k]), k
    assert f32(b["r_on_cp"]) == g["b_r_on_cp"]
    for k in ("t", "p", "qv", "nc"):
        assert np.array_equal(a[k], g["a_" + k]), k


def test_kat_b_inputs_formed_in_binary64_miss_the_native_digits(oracle_warm):
    """The control: the same run from inputs formed in binary64 and rounded (an ulp of binary32 apart in theta, exner, qv)
    ends 1.3e-5 from the native cloud water -- the distance that was wrongly booked on the oracle's arithmetic."""
    a, b = kc.kat_b(), kc.kat_b_native()
    for k in ("theta", "exner", "qv"):
        d = np.abs(a[k].astype(f32).astype(np.float64) / b[k].astype(np.float64) - 1.0)
        assert 0 < d.max() < 2.5e-7, (k, d.max())                    # at most two ulps of binary32, and not identical


@pytest.mark.slow
def test_kat_a_mixed_native_200_steps(oracle_mixed):
    st = kc.kat_a_native(True)
    ppt = None
    for _ in range(200):
        ppt, _, _, _ = oracle_mixed.column_step_p32n(st, 10.0)
    qi = float(st["qi"].astype(np.float64).sum())
    print("KAT-A mixed native, P32n oracle: sum qi %.7e, rain at call 200 %.7e" % (qi, float(ppt[0])))
    assert abs(qi / 3.76812e-4 - 1) < 4e-5                          # native; P64 gives 3.76939e-4 (3.4e-4 away); measured 2.1e-5
    assert abs(qi - 3.76812e-4) < 0.12 * abs(3.76939e-4 - 3.76812e-4)
    assert abs(float(ppt[0]) / 1.708911e-2 - 1) < 2e-6              # native rain precipitation at call 200 (measured 6.4e-7)


def test_p32n_step_stays_close_to_p64_after_one_call(oracle_warm):
    """One call from the same (binary32-representable) input: the two arithmetics differ at binary32 rounding level."""
    col = kc.kat_a(False)
    s32 = {k: np.ascontiguousarray(v.astype(f32)) for k, v in col.items()}
    s64 = {k: np.ascontiguousarray(s32[k].astype(np.float64)) for k in s32}
    oracle_warm.column_step_p32n(s32, 10.0)
    oracle_warm.column_step(s64, 10.0)
    for k, floor in (("qv", 1e-12), ("qc", 1e-9), ("qr", 1e-9), ("nr", 1e-3), ("t", 1.0)):
        e = np.abs(s32[k].astype(np.float64) - s64[k]) / np.maximum(np.abs(s64[k]), floor)
        assert float(e.max()) < 2e-3, (k, float(e.max()))
        assert float(np.median(e)) < 1e-5, (k, float(np.median(e)))
