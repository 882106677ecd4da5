"""On-device accuracy of the column kernel's math helpers (csrc/fastmath.h) through kidmp_math_probe.

The host build of the same header is covered by test_fastmath.py; this one runs the device instantiation
(v_rsq_f64 / v_log_f32 / v_exp_f32 seeds, v_frexp, v_ldexp) over the argument ranges the scheme produces and
compares with numpy's 80-bit long double.  Bounds in ulps of the correctly rounded result.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 400_000


def _ulp_err(got, want_ld):
    w = want_ld.astype(np.float64)
    u = np.abs(np.nextafter(w, np.inf) - w)
    return float(np.max(np.abs(got.astype(np.longdouble) - want_ld) / u))


@pytest.fixture(scope="module")
def rng():
    return np.random.default_rng(20240)


def _pos(rng, lo, hi):
    return 10.0 ** rng.uniform(lo, hi, N) * (1.0 + rng.uniform(0, 1, N))


def test_log_family(gpu_mixed, rng):
    x = _pos(rng, -45, 25)
    xl = x.astype(np.longdouble)
    assert _ulp_err(gpu_mixed.math_probe("log", x), np.log(xl)) <= 1.5
    assert _ulp_err(gpu_mixed.math_probe("log10", x), np.log10(xl)) <= 2.2


def test_exp_family(gpu_mixed, rng):
    a = rng.uniform(-100, 50, N)
    assert _ulp_err(gpu_mixed.math_probe("exp", a), np.exp(a.astype(np.longdouble))) <= 1.5
    b = rng.uniform(-40, 20, N)
    assert _ulp_err(gpu_mixed.math_probe("exp10", b), np.power(np.longdouble(10), b.astype(np.longdouble))) <= 1.5


def test_roots(gpu_mixed, rng):
    x = _pos(rng, -36, 36)
    xl = x.astype(np.longdouble)
    assert _ulp_err(gpu_mixed.math_probe("sqrt", x), np.sqrt(xl)) <= 1.0
    assert _ulp_err(gpu_mixed.math_probe("cbrt", x), np.cbrt(xl)) <= 1.0


def test_pow(gpu_mixed, rng):
    x = _pos(rng, -45, 25)
    y = rng.uniform(-4.2, 4.2, N)
    want = np.power(x.astype(np.longdouble), y.astype(np.longdouble))
    assert _ulp_err(gpu_mixed.math_probe("pow", x, y), want) <= 4.0


def test_division_and_reciprocal(gpu_mixed, rng):
    """The kernel's own fp64 division (one Newton step + residual correction on a 2^-24.4 seed) and reciprocal (one
    cubic step): <= 1 ulp, and the quotient equals IEEE division on (practically) every pair."""
    a, b = _pos(rng, -30, 30), _pos(rng, -30, 30)
    q = gpu_mixed.math_probe("div", a, b)
    assert _ulp_err(q, a.astype(np.longdouble) / b.astype(np.longdouble)) <= 1.0
    assert np.mean(q != gpu_mixed.math_probe("ieee_div", a, b)) < 1e-3
    assert _ulp_err(gpu_mixed.math_probe("rcp", a, b), 1.0 / b.astype(np.longdouble)) <= 1.0
    seed = gpu_mixed.math_probe("rcp_seed", a, b)
    assert float(np.max(np.abs(seed * b - 1.0))) < 2.0 ** -23          # what the step counts rest on
