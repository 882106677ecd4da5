"""Synthetic column batches of BASELINE.json's configs (recipes of SURVEY.md 8d).

All generators return a dict of float64 numpy arrays [ncol, nz] with the keys of
kid_amd.STATE_NAMES + ("p", "w", "dz"); nc/nwfa/nifa hold the non-aerosol
defaults of M:958-964 (what the build's KiD adapter feeds, decision U2).
"""
import numpy as np

import kat_cases as kc

NZ = 120
SEED = 20241008
KEYS = ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr", "nc", "nwfa", "nifa", "t", "p", "w", "dz")
KEYS_ARGS = KEYS   # positional order of mp_thompson's array dummies (M:1156-1157)


def _defaults(st, set_Nc=100.0):
    rho = 0.622 * st["p"] / (287.04 * st["t"] * (st["qv"] + 0.622))
    st["nc"] = set_Nc * 1e6 / rho
    st["nwfa"] = 11.1e6 / rho
    st["nifa"] = 0.5e6 * 0.01 / rho
    return st


def replicate(col, ncol):
    return {k: np.ascontiguousarray(np.broadcast_to(col[k], (ncol, col[k].shape[0])).copy()) for k in KEYS}


def warm_column_t0():
    """config 1 initial column: the KAT-B sounding of SURVEY 9h (dz=25 m, warm, static cloud/rain layer)."""
    c = kc.kat_b()
    nz = c["nz"]
    p = c["p0"] * c["exner"] ** (1.0 / c["r_on_cp"])
    st = dict(qv=c["qv"].copy(), qc=c["hydro"][0, 0, 0].copy(), qr=c["hydro"][0, 1, 0].copy(),
              nr=c["hydro"][1, 1, 0].copy(), qi=np.zeros(nz), ni=np.zeros(nz), qs=np.zeros(nz), qg=np.zeros(nz),
              t=c["theta"] * c["exner"], p=p, w=np.zeros(nz), dz=c["dz"].copy())
    _defaults(st)
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in st.items()}


def warm_column_t900():
    """config 2 base column (cloud + rain present): committed fixture, see golden/make_config2_column.py."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config2_column_t900.npz"))
    return {k: np.ascontiguousarray(f[k]) for k in KEYS}


def config2(ncol=10000):
    return replicate(warm_column_t900(), ncol)


def _perturb(base, ncol, rng, sigma=0.3, dT=1.5):
    st = replicate(base, ncol)
    for k in ("qc", "qi", "qr", "qs", "qg", "ni", "nr"):
        st[k] *= rng.lognormal(0.0, sigma, size=(ncol, 1))
    st["t"] += rng.uniform(-dT, dT, size=(ncol, 1))
    return _defaults(st)


def config3(ncol=100000, seed=SEED):
    """mixed-phase deep-convection columns: KAT-A mixed profile x seeded perturbations."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return _perturb(kc.kat_a(True), ncol, rng)


def config5(ncol=100000, seed=SEED):
    """sedimentation-heavy squall-line profile on the stretched grid (>=20 CFL substeps)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    z = np.cumsum(3.0 * 1.047 ** np.arange(NZ)) - 0.5 * 3.0 * 1.047 ** np.arange(NZ)
    base = kc.kat_c()
    m = (z > 4000) & (z < 9000)
    base["qs"][m] = 2e-3
    base["qi"][m] = 1e-4
    base["ni"][m] = 1e5
    return _perturb(base, ncol, rng)


def edge_cases():
    """Columns that exercise the branches the bulk configs rarely reach."""
    cols = []
    a = kc.kat_a(True)
    cols.append(a)                                                   # 0 plain mixed
    dry = {k: v.copy() for k, v in a.items()}                        # 1 no_micro: dry, nothing to do
    for k in ("qc", "qi", "qr", "qs", "qg", "ni", "nr"):
        dry[k][:] = 0.0
    dry["qv"] *= 0.3
    dry["qc"][5] = 5e-13                                             # below R1: must be zeroed even on early exit
    cols.append(dry)
    cold = {k: v.copy() for k, v in a.items()}                       # 2 surface below freezing, HGFR aloft
    cold["t"] -= 35.0
    cold["qv"] *= 0.08
    cold["qc"][95:110] = 3e-7
    cols.append(cold)
    tiny = {k: v.copy() for k, v in a.items()}                       # 3 values near the R1 thresholds
    for k in ("qc", "qi", "qr", "qs", "qg"):
        tiny[k] = np.where(tiny[k] > 0, 3e-12, 0.0)
    cols.append(tiny)
    heavy = kc.kat_c()                                               # 4 many substeps
    cols.append(heavy)
    warm = kc.kat_a(False)                                           # 5 warm only
    cols.append(warm)
    hm = {k: v.copy() for k, v in a.items()}                         # 6 Hallett-Mossop window, riming
    hm["t"] = np.where((hm["t"] < 273.15) & (hm["t"] > 263.0), hm["t"], hm["t"])
    hm["qc"] *= 4.0
    hm["qg"] *= 3.0
    cols.append(hm)
    subsat = {k: v.copy() for k, v in a.items()}                     # 7 strongly subsaturated: sublimation/evaporation
    subsat["qv"] *= 0.4
    cols.append(subsat)
    snowy = {k: v.copy() for k, v in cold.items()}                   # 8 frozen precipitation reaching a cold surface
    snowy["qs"][:20] = 1.5e-3
    snowy["qg"][:20] = 2.5e-3
    snowy["qi"][:20] = 2e-4
    snowy["ni"][:20] = 3e4
    cols.append(snowy)
    out = {k: np.ascontiguousarray(np.stack([c[k] for c in cols])) for k in KEYS}
    return _defaults(out) if False else out
