"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; kid_amd/ never does.  See oracle/thompson_oracle.h for what the
oracle restates (reference file:line) and how its parity is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KIDMP_ORACLE_LIB: another build of the same checker (the sanitizer build oracle/_asan/..., tests/test_sanitizers.py)
_LIB = os.environ.get("KIDMP_ORACLE_LIB") or os.path.join(_HERE, "libthompson_oracle.so")
_CACHE_DIR = os.path.join(_HERE, "_cache")
NRATES = 36
RATE_NAMES = (
    "pri_inu pri_ide prs_ide prs_sde prg_gde pri_wfz prs_scw prg_scw prg_gcw "
    "pri_ihm pri_rfz prs_iau prs_sci pri_rci pni_inu pni_ihm pni_wfz pni_rfz "
    "pni_ide pni_iau pni_sci pni_rci prr_sml prr_gml pnr_rcs pnr_rcg pnr_rci "
    "pnr_sml pnr_gml pnr_rfz prr_wau prr_rcw prv_rev pnr_wau pnr_rev pnr_rcr"
).split()

_dp = C.POINTER(C.c_double)


def build(force=False):
    """Compile oracle/libthompson_oracle.so with gcc (building the checker)."""
    srcs = [os.path.join(_HERE, f) for f in (
        "thompson_oracle_init.c", "thompson_oracle_column.c", "thompson_oracle_p32n.c",
        "thompson_oracle.h", "thompson_oracle_internal.h", "Makefile")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.th_oracle_create.restype = C.c_void_p
        L.th_oracle_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_char_p]
        L.th_oracle_destroy.argtypes = [C.c_void_p]
        L.th_oracle_set_aerosol_aware.argtypes = [C.c_void_p, C.c_int]
        L.th_oracle_mp_thompson.restype = C.c_int
        L.th_oracle_mp_thompson.argtypes = [C.c_void_p] + [_dp] * 16 + [
            C.c_int, C.c_double, _dp, C.POINTER(C.c_int)]
        L.th_oracle_batch.restype = C.c_int
        L.th_oracle_batch.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double] + [_dp] * 16 + [C.c_int]
        L.th_oracle_batch_ex.restype = C.c_int
        L.th_oracle_batch_ex.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double] + [_dp] * 16 + [
            C.c_int, C.POINTER(C.c_int)]
        L.th_oracle_batch_force.restype = C.c_int
        L.th_oracle_batch_force.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_double] + [_dp] * 16 + [
            C.c_int, C.POINTER(C.c_int), C.c_int]
        _fp = C.POINTER(C.c_float)
        L.th_oracle_batch_force_p32n.restype = C.c_int
        L.th_oracle_batch_force_p32n.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_float] + [_fp] * 16 + [
            C.c_int, C.POINTER(C.c_int), C.c_int]
        L.th_oracle_mp_thompson_force_p32n.restype = C.c_int
        L.th_oracle_mp_thompson_force_p32n.argtypes = [C.c_void_p] + [_fp] * 16 + [
            C.c_int, C.c_float, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        L.th_oracle_kid_interface_p32n.restype = C.c_int
        L.th_oracle_kid_interface_p32n.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float] + [_fp] * 15
        for f in ("th_oracle_view_const", "th_oracle_view_const_p32n"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.th_oracle_calc_effectRad.restype = None
        L.th_oracle_calc_effectRad.argtypes = [C.c_void_p, C.c_int] + [_dp] * 11
        L.th_oracle_default_aerosols.argtypes = [C.c_void_p, C.c_int] + [_dp] * 6
        L.th_oracle_kid_interface.restype = C.c_int
        L.th_oracle_kid_interface.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double,
                                              C.c_double, C.c_double] + [_dp] * 15
        L.th_oracle_const.restype = _dp
        L.th_oracle_const.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.th_oracle_table.restype = _dp
        L.th_oracle_table.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.th_oracle_int.restype = C.c_int
        L.th_oracle_int.argtypes = [C.c_void_p, C.c_char_p]
        for f in ("th_oracle_rslf", "th_oracle_rsif", "th_oracle_gammp"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double, C.c_double]
        L.th_oracle_gammln.restype = C.c_double
        L.th_oracle_gammln.argtypes = [C.c_double]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


STATE = ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr", "nc", "nwfa", "nifa", "t")
FORCING = ("p", "w", "dz")


class Oracle:
    """thompson_init + mp_thompson of the reference, restated on the CPU."""

    def __init__(self, iiwarm=False, set_Nc=100.0, l_sediment=True, nthreads=None, cache=True, aerosol_aware=False):
        nthreads = nthreads or min(os.cpu_count() or 1, 16)
        path = None
        if cache and not iiwarm:
            os.makedirs(_CACHE_DIR, exist_ok=True)
            path = os.path.join(_CACHE_DIR, "tables_p64_nc%g.bin" % set_Nc).encode()
        self.iiwarm = bool(iiwarm)
        self.nthreads = nthreads
        self._h = lib().th_oracle_create(int(iiwarm), float(set_Nc), int(l_sediment), nthreads, path)
        if not self._h:
            raise MemoryError("th_oracle_create failed")
        if aerosol_aware:
            self.set_aerosol_aware(True)

    def set_aerosol_aware(self, flag):
        """is_aerosol_aware of M:28 (call between steps only)."""
        lib().th_oracle_set_aerosol_aware(self._h, int(bool(flag)))

    def close(self):
        if getattr(self, "_h", None):
            lib().th_oracle_destroy(self._h)
            self._h = None

    __del__ = close

    # -- one column --------------------------------------------------
    def column_step(self, st, dt, want_rates=False):
        """st: dict of 1-D float64 arrays (STATE + FORCING), updated in place.
        Returns (ppt[4] = rain,snow,graupel,ice, rates or None, nstep[4], no_micro)."""
        nz = st["qv"].shape[0]
        for k in STATE + FORCING:
            assert st[k].dtype == np.float64 and st[k].flags.c_contiguous and st[k].shape == (nz,)
        ppt = np.zeros(4)
        rates = np.zeros((NRATES, nz)) if want_rates else None
        nstep = (C.c_int * 4)()
        rc = lib().th_oracle_mp_thompson(
            self._h, *[_p(st[k]) for k in STATE], *[_p(st[k]) for k in FORCING], _p(ppt),
            nz, float(dt), _p(rates) if want_rates else None, nstep)
        if rc < 0:
            raise MemoryError
        return ppt, rates, list(nstep), bool(rc)

    # -- batch, k-fastest [ncol, nz] ----------------------------------
    def batch_step(self, st, dt, nthreads=None, want_illcond=False, force=0):
        """In-place step of [ncol, nz] arrays; returns ppt [ncol, 4] (and, if asked, the int32
        [ncol, nz] ill-conditioning flags of th_oracle_mp_thompson_ex).  force = 1 / 2 takes the
        residue-decided tests of block Q (M:3587, M:3596) as true / false at the flagged levels."""
        ncol, nz = st["qv"].shape
        for k in STATE + FORCING:
            assert st[k].dtype == np.float64 and st[k].flags.c_contiguous and st[k].shape == (ncol, nz), k
        ppt = np.zeros((ncol, 4))
        flags = np.zeros((ncol, nz), dtype=np.int32) if want_illcond else None
        lib().th_oracle_batch_force(self._h, ncol, nz, float(dt), *[_p(st[k]) for k in STATE],
                                    *[_p(st[k]) for k in FORCING], _p(ppt), nthreads or self.nthreads,
                                    flags.ctypes.data_as(C.POINTER(C.c_int)) if want_illcond else None, int(force))
        return (ppt, flags) if want_illcond else ppt

    # -- P32n: the reference's native arithmetic (REAL = binary32) ------------
    def batch_step_p32n(self, st, dt, nthreads=None):
        """In-place step of float32 [ncol, nz] arrays in the reference's native arithmetic; returns ppt float32 [ncol, 4]."""
        ncol, nz = st["qv"].shape
        fp = C.POINTER(C.c_float)
        for k in STATE + FORCING:
            assert st[k].dtype == np.float32 and st[k].flags.c_contiguous and st[k].shape == (ncol, nz), k
        ppt = np.zeros((ncol, 4), dtype=np.float32)
        lib().th_oracle_batch_force_p32n(self._h, ncol, nz, float(dt), *[st[k].ctypes.data_as(fp) for k in STATE],
                                         *[st[k].ctypes.data_as(fp) for k in FORCING], ppt.ctypes.data_as(fp),
                                         nthreads or self.nthreads, None, 0)
        return ppt

    def column_step_p32n(self, st, dt, want_rates=False):
        """One float32 column in place; returns (ppt[4] float32, rates float64 [36, nz] or None, nstep[4], no_micro)."""
        nz = st["qv"].shape[0]
        fp = C.POINTER(C.c_float)
        for k in STATE + FORCING:
            assert st[k].dtype == np.float32 and st[k].flags.c_contiguous and st[k].shape == (nz,), k
        ppt = np.zeros(4, dtype=np.float32)
        rates = np.zeros((NRATES, nz)) if want_rates else None
        nstep = (C.c_int * 4)()
        rc = lib().th_oracle_mp_thompson_force_p32n(
            self._h, *[st[k].ctypes.data_as(fp) for k in STATE], *[st[k].ctypes.data_as(fp) for k in FORCING],
            ppt.ctypes.data_as(fp), nz, float(dt), _p(rates) if want_rates else None, nstep, None, 0)
        if rc < 0:
            raise MemoryError
        return ppt, rates, list(nstep), bool(rc)

    def kid_interface_p32n(self, nz, nx, dt, p0, r_on_cp, theta, dtheta_adv, dtheta_div, exner, dz,
                           qv, dqv_adv, dqv_div, hydro, dhydro_adv, dhydro_div):
        """The KiD adapter (W:28-310) with 4-byte REALs throughout, as KiD's default build runs it."""
        fp = C.POINTER(C.c_float)
        dth = np.zeros(nz * nx, dtype=np.float32); dqv = np.zeros(nz * nx, dtype=np.float32)
        dhy = np.zeros(nz * nx * 10, dtype=np.float32); ppt = np.zeros(4 * nx, dtype=np.float32)
        args = [np.ascontiguousarray(a, dtype=np.float32).ravel() for a in (
            theta, dtheta_adv, dtheta_div, exner, dz, qv, dqv_adv, dqv_div, hydro, dhydro_adv, dhydro_div)]
        lib().th_oracle_kid_interface_p32n(self._h, nz, nx, float(dt), float(p0), float(r_on_cp),
                                           *[a.ctypes.data_as(fp) for a in args], dth.ctypes.data_as(fp),
                                           dqv.ctypes.data_as(fp), dhy.ctypes.data_as(fp), ppt.ctypes.data_as(fp))
        return dth, dqv, dhy, ppt.reshape(4, nx)

    def view_const(self, name, idx=0, p32n=False):
        f = lib().th_oracle_view_const_p32n if p32n else lib().th_oracle_view_const
        return f(self._h, name.encode(), int(idx))

    def calc_effectRad(self, st, preset=(2.49e-6, 4.99e-6, 9.99e-6)):
        """calc_effectRad (M:4834-4935) on [ncol, nz] or [nz] arrays; returns (re_qc, re_qi, re_qs) started from the
        driver's presets (M:1111-1113)."""
        shp = st["qv"].shape
        nz = shp[-1]
        a = {k: np.ascontiguousarray(st[k].reshape(-1, nz)) for k in ("t", "p", "qv", "qc", "nc", "qi", "ni", "qs")}
        out = [np.full_like(a["t"], v) for v in preset]
        for i in range(a["t"].shape[0]):
            lib().th_oracle_calc_effectRad(self._h, nz, *[_p(a[k][i]) for k in ("t", "p", "qv", "qc", "nc", "qi", "ni", "qs")],
                                           *[_p(o[i]) for o in out])
        return tuple(o.reshape(shp) for o in out)

    def default_aerosols(self, qv, t, p):
        nz = qv.shape[-1]
        qv2, t2, p2 = (np.ascontiguousarray(a.reshape(-1, nz)) for a in (qv, t, p))
        nc = np.empty_like(qv2); nwfa = np.empty_like(qv2); nifa = np.empty_like(qv2)
        for i in range(qv2.shape[0]):
            lib().th_oracle_default_aerosols(self._h, nz, _p(qv2[i]), _p(t2[i]), _p(p2[i]),
                                             _p(nc[i]), _p(nwfa[i]), _p(nifa[i]))
        return nc.reshape(qv.shape), nwfa.reshape(qv.shape), nifa.reshape(qv.shape)

    # -- KiD adapter ----------------------------------------------------
    def kid_interface(self, nz, nx, dt, p0, r_on_cp, theta, dtheta_adv, dtheta_div, exner, dz,
                      qv, dqv_adv, dqv_div, hydro, dhydro_adv, dhydro_div):
        """Arrays in Fortran order flattened: theta[k + nz*i]; hydro[k + nz*(i + nx*(ih + 5*imom))]."""
        dth = np.zeros(nz * nx); dqv = np.zeros(nz * nx); dhy = np.zeros(nz * nx * 10)
        ppt = np.zeros(4 * nx)
        args = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in (
            theta, dtheta_adv, dtheta_div, exner, dz, qv, dqv_adv, dqv_div, hydro, dhydro_adv, dhydro_div)]
        lib().th_oracle_kid_interface(self._h, nz, nx, float(dt), float(p0), float(r_on_cp),
                                      *[_p(a) for a in args], _p(dth), _p(dqv), _p(dhy), _p(ppt))
        return dth, dqv, dhy, ppt.reshape(4, nx)

    # -- introspection ----------------------------------------------------
    def const(self, name):
        n = C.c_int()
        p = lib().th_oracle_const(self._h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def table(self, name):
        """Returns the table as a numpy array indexed like the Fortran (i,j,k,m), 0-based."""
        nd = C.c_int(); dims = (C.c_int * 4)()
        p = lib().th_oracle_table(self._h, name.encode(), C.byref(nd), dims)
        if not p:
            raise KeyError(name)
        shape = tuple(dims[i] for i in range(nd.value))
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(shape, order="F")

    def integer(self, name):
        return lib().th_oracle_int(self._h, name.encode())


def rslf(p, t):
    return lib().th_oracle_rslf(p, t)


def rsif(p, t):
    return lib().th_oracle_rsif(p, t)


def gammln(x):
    return lib().th_oracle_gammln(x)


def gammp(a, x):
    return lib().th_oracle_gammp(a, x)
