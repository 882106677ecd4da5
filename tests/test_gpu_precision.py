"""The binary32 builds of the column kernel (-m gpu): "p32n" = the reference's native arithmetic (REAL = binary32
state and work variables, DOUBLE PRECISION = binary64 rates, M:1168-1253) and "f32" = everything binary32.

Oracle: the same split compiled into the CPU oracle (oracle/thompson_oracle_p32n.c), which is pinned on the native
known answers of the survey (tests/test_oracle_p32n.py).  Two binary32 implementations agree only to binary32
rounding amplified by the scheme (their libm differs: glibc powf/expf vs the fp32 special-function units), so the
bounds are statistical: medians at a few binary32 ulps, high percentiles at 1e-4, with the same floors as the P64
metric.  P32n-vs-P64 and f32-vs-P64 differences are reported by tools/precision_sweep.py (BASELINE config 5)."""
import numpy as np
import pytest

import cases
import kat_cases as kc
from parity import FLOORS, OUT

pytestmark = pytest.mark.gpu
f32 = np.float32


def _to32(st):
    return {k: np.ascontiguousarray(v.astype(f32)) for k, v in st.items()}


def _err(a, b, k):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e4 * FLOORS[k])          # floors: 1e-8 kg/kg, 1e-2 /kg


def _stats(got, ref):
    e = np.concatenate([_err(got[k], ref[k], k).ravel() for k in OUT])
    return float(np.median(e)), float(np.quantile(e, 0.99)), float(np.quantile(e, 0.9999)), float(e.max())


@pytest.mark.parametrize("name,iiwarm", [("config2", True), ("config3", False), ("config5", False)])
def test_p32n_kernel_against_the_p32n_oracle(name, iiwarm, gpu_warm, gpu_mixed, oracle_warm, oracle_mixed):
    m, o = (gpu_warm, oracle_warm) if iiwarm else (gpu_mixed, oracle_mixed)
    st = _to32(getattr(cases, name)(256))
    if name == "config2":
        st["qr"] *= np.linspace(0.5, 1.5, 256, dtype=f32)[:, None]
    ref = {k: v.copy() for k, v in st.items()}
    rppt = o.batch_step_p32n(ref, 10.0)
    got = {k: v.copy() for k, v in st.items()}
    gppt, _, _ = m.batch_step32_host(got, 10.0, arith="p32n")
    med, q99, q9999, mx = _stats(got, ref)
    print(name, "p32n gpu vs p32n oracle: median %.1e q99 %.1e q99.99 %.1e max %.1e" % (med, q99, q9999, mx))
    assert med < 3e-7 and q99 < 1e-4, (med, q99, q9999, mx)
    assert all(np.isfinite(got[k]).all() for k in OUT)
    pe = np.abs(gppt.astype(np.float64) - rppt) / np.maximum(np.abs(rppt), 1e-8)
    assert float(pe.max()) < 1e-4, float(pe.max())


@pytest.mark.parametrize("arith,bound", [("p32n", 2e-3), ("f32", 2e-3)])
def test_binary32_builds_stay_near_the_p64_result(arith, bound, gpu_mixed, oracle_mixed):
    """One step from identical (binary32-representable) inputs: both builds differ from P64 at binary32 rounding
    level (99th percentile), and every invariant of the step holds."""
    st32 = _to32(cases.config5(512))
    st64 = {k: np.ascontiguousarray(v.astype(np.float64)) for k, v in st32.items()}
    oracle_mixed.batch_step(st64, 10.0)
    got = {k: v.copy() for k, v in st32.items()}
    ppt, rates, nstep = gpu_mixed.batch_step32_host(got, 10.0, arith=arith, want_rates=True, want_nstep=True)
    med, q99, q9999, mx = _stats(got, st64)
    print(arith, "vs P64: median %.1e q99 %.1e q99.99 %.1e max %.1e" % (med, q99, q9999, mx))
    assert med < 1e-6 and q99 < bound, (med, q99)
    assert (got["qv"] >= f32(1e-10)).all() and (ppt >= 0).all() and np.isfinite(rates).all()
    for q in ("qc", "qi", "qr", "qs", "qg"):
        assert ((got[q] == 0) | (got[q] > f32(1e-12))).all(), q
    assert np.median(nstep[:, 0]) >= 20                                   # the substep count is that of the P64 run


def test_kat_b_native_through_the_p32n_kernel(gpu_warm):
    """the reference AS SHIPPED ends KAT-B (360 warm steps through the adapter) at sum(qc) 2.218541e-2,
    sum(qr) 2.693803e-3, sum(nr) 1.060634e6; its P64 build at 2.218719e-2, 2.694135e-3, 1.060568e6.  The p32n kernel,
    driven like the adapter drives it (state + dt*tendency in binary32), lands near the native numbers, not the P64 ones."""
    c = kc.kat_b_native()                      # inputs formed in binary32, as the survey's native probe forms them
    nz, dt = c["nz"], f32(c["dt"])
    theta, qv = c["theta"].copy(), c["qv"].copy()
    qc, qr, nr = (c["hydro"][0, 0, 0].copy(), c["hydro"][0, 1, 0].copy(), c["hydro"][1, 1, 0].copy())
    exner, dz = c["exner"], c["dz"]
    m = kc._libm32()                           # p1d = p0*exner**(1./r_on_cp), W:61: REAL**REAL = powf
    ex = float(f32(1.0) / f32(c["r_on_cp"]))
    p = (f32(c["p0"]) * np.array([m.powf(float(e), ex) for e in exner], dtype=f32)).astype(f32)
    z = np.zeros(nz, dtype=f32)
    for _ in range(360):
        t = (theta * exner).astype(f32)
        rho = (f32(0.622) * p / (f32(287.04) * t * (qv + f32(0.622)))).astype(f32)
        st = dict(qv=qv.copy(), qc=qc.copy(), qi=z.copy(), qr=qr.copy(), qs=z.copy(), qg=z.copy(), ni=z.copy(), nr=nr.copy(),
                  nc=(f32(1e8) / rho).astype(f32), nwfa=(f32(11.1e6) / rho).astype(f32), nifa=(f32(5e3) / rho).astype(f32),
                  t=t, p=p, w=z.copy(), dz=dz)
        st = {k: np.ascontiguousarray(v[None, :]) for k, v in st.items()}
        gpu_warm.batch_step32_host(st, float(dt), arith="p32n")
        # the adapter's tendencies (W:198-245) and KiD's update, in binary32
        theta = (theta + dt * ((st["t"][0] / exner - theta) / dt)).astype(f32)
        qv = (qv + dt * ((st["qv"][0] - qv) / dt)).astype(f32)
        qc = (qc + dt * ((st["qc"][0] - qc) / dt)).astype(f32)
        qr = (qr + dt * ((st["qr"][0] - qr) / dt)).astype(f32)
        nr = (nr + dt * ((st["nr"][0] - nr) / dt)).astype(f32)
    got = [float(a.astype(np.float64).sum()) for a in (qc, qr, nr)]
    native, p64 = [2.218541e-2, 2.693803e-3, 1.060634e6], [2.218719e-2, 2.694135e-3, 1.060568e6]
    print("KAT-B p32n kernel:", got)
    for g, n, pp in zip(got, native, p64):
        assert abs(g / n - 1) < 2.5e-5, (got, native)     # measured 4e-6, 1.5e-5, 1.1e-5 (binary32 special-function units, not libm)
        assert abs(g - n) < 0.5 * abs(pp - n), (g, n, pp)
