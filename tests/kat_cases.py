"""Input soundings of the survey's known-answer probes.

These are the exact recipes the survey ran through the reference Fortran (P64
build); the expected numbers quoted in tests/test_oracle_kat.py are the
reference's own outputs as recorded there (6-7 significant digits).
"""
import numpy as np

NZ = 120


def _thermo(z):
    T = np.maximum(210.0, 300.0 - 6.5e-3 * z)
    p = 1e5 * (1.0 - 2.2557e-5 * z) ** 5.2559
    es = 611.2 * np.exp(17.67 * (T - 273.15) / (T - 29.65))
    qsat = 0.622 * es / (p - es)
    return T, p, qsat


def _finish(st, z, dz):
    T, p, qv = st["t"], st["p"], st["qv"]
    rho = 0.622 * p / (287.04 * T * (qv + 0.622))
    st["nc"] = 1e8 / rho
    st["nwfa"] = 11.1e6 / rho
    st["nifa"] = 5e3 / rho
    st["w"] = np.zeros(NZ)
    st["dz"] = np.ascontiguousarray(dz, dtype=np.float64)
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in st.items()}


def kat_a(mixed):
    """KAT-A: direct mp_thompson, uniform 125 m grid."""
    dz = np.full(NZ, 125.0)
    z = (np.arange(1, NZ + 1) - 0.5) * 125.0
    T, p, qsat = _thermo(z)
    st = {k: np.zeros(NZ) for k in ("qc", "qi", "qr", "qs", "qg", "ni", "nr")}
    st["t"], st["p"] = T, p
    qv = 0.7 * qsat
    m = (z > 1000) & (z < 4000)
    qv[m] = 1.02 * qsat[m]
    st["qc"][m] = 1e-3
    st["qr"][m] = 5e-4
    st["nr"][m] = 5e3
    if mixed:
        m = (z > 4000) & (z < 11000)
        qv[m] = qsat[m]
        st["qc"][m] = 2e-4
        st["qi"][m] = 1e-4
        st["ni"][m] = 1e5
        st["qs"][m] = 1e-3
        st["qg"][m] = 2e-3
        st["qr"][m] = 1e-4
        st["nr"][m] = 1e3
    st["qv"] = qv
    return _finish(st, z, dz)


def kat_c():
    """KAT-C: config-5 column (stretched grid, >=20 CFL substeps)."""
    dz = 3.0 * 1.047 ** np.arange(NZ)
    ztop = np.cumsum(dz)
    z = ztop - 0.5 * dz
    T, p, qsat = _thermo(z)
    st = {k: np.zeros(NZ) for k in ("qc", "qi", "qr", "qs", "qg", "ni", "nr")}
    st["t"], st["p"] = T, p
    st["qv"] = 0.9 * qsat
    m = z < 4000
    st["qr"][m] = 5e-3
    st["nr"][m] = 2e3
    st["qg"][m] = 8e-3
    m = (z > 4000) & (z < 9000)
    st["qg"][m] = 4e-3
    return _finish(st, z, dz)


def kat_b():
    """KAT-B: KiD adapter inputs (warm, nx=1).  Returns dict of Fortran-order arrays."""
    nz = NZ
    dz = np.full(nz, 25.0)
    z = (np.arange(1, nz + 1) - 0.5) * 25.0
    p = 1e5 * (1.0 - 2.2557e-5 * z) ** 5.2559
    r_on_cp = 287.058 / 1005.0
    exner = (p / 1e5) ** r_on_cp
    T = 297.0 - 6.5e-3 * z
    theta = T / exner
    qv = 0.015 - 0.004 * z / 3000.0
    hydro = np.zeros((2, 5, 1, nz))          # [imom, ih, i, k] == Fortran (k,i,ih,imom)
    m = (z > 800) & (z < 2000)
    hydro[0, 0, 0, m] = 8e-4
    hydro[0, 1, 0, m] = 3e-4
    hydro[1, 1, 0, m] = 2e4
    return dict(nz=nz, nx=1, dt=10.0, p0=1e5, r_on_cp=r_on_cp, theta=theta, exner=exner, dz=dz,
                qv=qv, hydro=hydro)


# ---- the same recipes as a default-REAL (binary32) Fortran driver forms them ----
# The survey's NATIVE numbers ("native P32n build") come from a build in which the probe driver itself is
# default REAL: z, p, exner, T, theta, qv ... are formed in binary32 arithmetic, with the binary32 libm (powf, expf: the same
# glibc functions a flang-built program calls).  Forming them in binary64 and rounding afterwards differs by an ulp of
# binary32 in theta, exner and qv at some levels -- and 360 coupled steps turn that into 1.3e-5 of the cloud water, which is
# what separated the P32n oracle from the recorded native digits until round 4.
def _libm32():
    import ctypes
    import ctypes.util
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for name, nargs in (("powf", 2), ("expf", 1)):
        f = getattr(m, name)
        f.restype = ctypes.c_float
        f.argtypes = [ctypes.c_float] * nargs
    return m


def _f32map(fn, *arrs):
    f32 = np.float32
    return np.array([f32(fn(*[float(a[i]) for a in arrs])) for i in range(len(arrs[0]))], dtype=f32)


def kat_b_native():
    """KAT-B inputs in binary32 arithmetic (see above); same keys as kat_b(), arrays binary32."""
    f32 = np.float32
    m = _libm32()
    nz = NZ
    dz = np.full(nz, 25.0, dtype=f32)
    z = ((np.arange(1, nz + 1).astype(f32) - f32(0.5)) * f32(25.0)).astype(f32)
    base = (f32(1.0) - f32(2.2557e-5) * z).astype(f32)
    p = (f32(1e5) * _f32map(lambda b: m.powf(b, float(f32(5.2559))), base)).astype(f32)
    r_on_cp = f32(287.058) / f32(1005.0)
    exner = _f32map(lambda q: m.powf(q, float(r_on_cp)), (p / f32(1e5)).astype(f32))
    T = (f32(297.0) - f32(6.5e-3) * z).astype(f32)
    theta = (T / exner).astype(f32)
    qv = (f32(0.015) - f32(0.004) * z / f32(3000.0)).astype(f32)
    hydro = np.zeros((2, 5, 1, nz), dtype=f32)
    mk = (z > 800) & (z < 2000)
    hydro[0, 0, 0, mk] = f32(8e-4)
    hydro[0, 1, 0, mk] = f32(3e-4)
    hydro[1, 1, 0, mk] = f32(2e4)
    return dict(nz=nz, nx=1, dt=10.0, p0=float(f32(1e5)), r_on_cp=float(r_on_cp), theta=theta, exner=exner, dz=dz, qv=qv, hydro=hydro)


def kat_a_native(mixed):
    """KAT-A inputs in binary32 arithmetic; same keys as kat_a(), arrays binary32."""
    f32 = np.float32
    m = _libm32()
    dz = np.full(NZ, 125.0, dtype=f32)
    z = ((np.arange(1, NZ + 1).astype(f32) - f32(0.5)) * f32(125.0)).astype(f32)
    T = np.maximum(f32(210.0), (f32(300.0) - f32(6.5e-3) * z).astype(f32)).astype(f32)
    p = (f32(1e5) * _f32map(lambda b: m.powf(b, float(f32(5.2559))), (f32(1.0) - f32(2.2557e-5) * z).astype(f32))).astype(f32)
    arg = (f32(17.67) * (T - f32(273.15)).astype(f32) / (T - f32(29.65)).astype(f32)).astype(f32)
    es = (f32(611.2) * _f32map(m.expf, arg)).astype(f32)
    qsat = (f32(0.622) * es / (p - es).astype(f32)).astype(f32)
    st = {k: np.zeros(NZ, dtype=f32) for k in ("qc", "qi", "qr", "qs", "qg", "ni", "nr")}
    st["t"], st["p"] = T, p
    qv = (f32(0.7) * qsat).astype(f32)
    mk = (z > 1000) & (z < 4000)
    qv[mk] = (f32(1.02) * qsat[mk]).astype(f32)
    st["qc"][mk] = f32(1e-3); st["qr"][mk] = f32(5e-4); st["nr"][mk] = f32(5e3)
    if mixed:
        mk = (z > 4000) & (z < 11000)
        qv[mk] = qsat[mk]
        st["qc"][mk] = f32(2e-4); st["qi"][mk] = f32(1e-4); st["ni"][mk] = f32(1e5)
        st["qs"][mk] = f32(1e-3); st["qg"][mk] = f32(2e-3); st["qr"][mk] = f32(1e-4); st["nr"][mk] = f32(1e3)
    st["qv"] = qv
    rho = (f32(0.622) * p / (f32(287.04) * T * (qv + f32(0.622)).astype(f32)).astype(f32)).astype(f32)
    st["nc"] = (f32(1e8) / rho).astype(f32); st["nwfa"] = (f32(11.1e6) / rho).astype(f32); st["nifa"] = (f32(5e3) / rho).astype(f32)
    st["w"] = np.zeros(NZ, dtype=f32)
    st["dz"] = dz
    return {k: np.ascontiguousarray(v, dtype=f32) for k, v in st.items()}
