"""Host side of the MI355X Thompson-09n column solver.

Mirrors the reference's operator interface for the hot path
(/root/reference/module_mp_thompson09n.f90, "M:"):

    thompson_init()                 M:374      -> thompson_init(...) / ThompsonMP(...)
    mp_thompson(qv1d, ..., dt)      M:1156     -> mp_thompson(...) / ThompsonMP.mp_thompson
    do i=1,nx ... (KiD adapter)     W:54-246   -> ThompsonMP.batch_step (one launch)

Everything numerical happens in kid_amd/libkidmp.so (HIP kernels, C ABI of
include/kidmp.h).  There is deliberately NO CPU fallback: if the library or a
gfx950 device is missing, calls raise KidmpError.  torch is used only as the
owner of device memory and streams.
"""
import ctypes as C
import os

import numpy as np

STATE_NAMES = ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr", "nc", "nwfa", "nifa", "t")   # INOUT, M:1168-1170
FORCING_NAMES = ("p", "w", "dz")                                                               # IN, M:1171
RATE_NAMES = (
    "pri_inu pri_ide prs_ide prs_sde prg_gde pri_wfz prs_scw prg_scw prg_gcw "
    "pri_ihm pri_rfz prs_iau prs_sci pri_rci pni_inu pni_ihm pni_wfz pni_rfz "
    "pni_ide pni_iau pni_sci pni_rci prr_sml prr_gml pnr_rcs pnr_rcg pnr_rci "
    "pnr_sml pnr_gml pnr_rfz prr_wau prr_rcw prv_rev pnr_wau pnr_rev pnr_rcr"
).split()                                                                                      # M:2967-3119
NRATES = 36
MAX_NZ = 256
PPT_LIMBS = 24            # KIDMP_PPT_LIMBS: exact precipitation accumulators (include/kidmp.h)

_HERE = os.path.dirname(os.path.abspath(__file__))


class KidmpError(RuntimeError):
    pass


class _Cfg(C.Structure):
    _fields_ = [("iiwarm", C.c_int32), ("l_sediment", C.c_int32), ("set_Nc", C.c_double),
                ("device", C.c_int32), ("is_aerosol_aware", C.c_int32)]


def lib_path():
    """The in-tree build.  No environment override: a profiling or A/B build is selected explicitly with
    load_library(path) (bench.py --lib) before the first context is made."""
    return os.path.join(_HERE, "libkidmp.so")


_lib = None
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p


def load_library(path=None):
    """dlopen kid_amd/libkidmp.so (or an explicitly named build of it) and declare the C ABI (include/kidmp.h)."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.abspath(path) != _lib._name:
            raise KidmpError("load_library: %s is already loaded" % _lib._name)
        return _lib
    path = os.path.abspath(path) if path else lib_path()
    # torch ships its own HIP runtime: when this process is going to use torch (the device entries of this mirror
    # take torch tensors) it must be loaded FIRST, so that libkidmp.so binds to the same libamdhip64 -- two HIP
    # runtimes in one process leave the second one without a device ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise KidmpError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = C.CDLL(path)
    L.kidmp_init.restype = C.c_int
    L.kidmp_init.argtypes = [C.POINTER(_Cfg), C.POINTER(_vp)]
    L.kidmp_finalize.restype = None
    L.kidmp_finalize.argtypes = [_vp]
    L.kidmp_last_error.restype = C.c_char_p
    L.kidmp_last_error.argtypes = [_vp]
    L.kidmp_column_step.restype = C.c_int
    L.kidmp_column_step.argtypes = [_vp, C.c_int32, C.c_double] + [_dp] * 16
    L.kidmp_batch_step_host.restype = C.c_int
    L.kidmp_batch_step_host.argtypes = [_vp, C.c_int64, C.c_int32, C.c_double] + [_dp] * 17
    L.kidmp_batch_step_device.restype = C.c_int
    L.kidmp_batch_step_device.argtypes = [_vp, C.c_int64, C.c_int32, C.c_double] + [_vp] * 18 + [_vp]
    _fpp = C.POINTER(C.c_float)
    L.kidmp32_batch_step_host.restype = C.c_int
    L.kidmp32_batch_step_host.argtypes = [_vp, C.c_int64, C.c_int32, C.c_float] + [_fpp] * 16 + [_dp, C.POINTER(C.c_int32), C.c_int32]
    L.kidmp32_batch_step_device.restype = C.c_int
    L.kidmp32_batch_step_device.argtypes = [_vp, C.c_int64, C.c_int32, C.c_float] + [_vp] * 18 + [C.c_int32, _vp]
    L.kidmp32_column_step.restype = C.c_int
    L.kidmp32_column_step.argtypes = [_vp, C.c_int32, C.c_float] + [_fpp] * 16 + [C.c_int32]
    L.kidmp_default_aerosols_device.restype = C.c_int
    L.kidmp_default_aerosols_device.argtypes = [_vp, C.c_int64] + [_vp] * 6 + [_vp]
    L.kidmp_reduce_ppt_device.restype = C.c_int
    L.kidmp_reduce_ppt_device.argtypes = [_vp, C.c_int64, _vp, _vp, _vp]
    L.kidmp_reduce_rates_device.restype = C.c_int
    L.kidmp_reduce_rates_device.argtypes = [_vp, C.c_int64, C.c_int32, _vp, _vp, _vp]
    L.kidmp_sanity_device.restype = C.c_int
    L.kidmp_sanity_device.argtypes = [_vp, C.c_int64] + [_vp] * 9 + [_vp]
    L.kidmp_effective_radii_device.restype = C.c_int
    L.kidmp_effective_radii_device.argtypes = [_vp, C.c_int64] + [_vp] * 11 + [_vp]
    L.kidmp_kernel_fingerprint.restype = C.c_char_p
    L.kidmp_kernel_fingerprint.argtypes = [_vp]
    L.kidmp32_kernel_fingerprint.restype = C.c_char_p
    L.kidmp32_kernel_fingerprint.argtypes = [_vp, C.c_int32]
    L.kidmp_reduce_ppt_exact_device.restype = C.c_int
    L.kidmp_reduce_ppt_exact_device.argtypes = [_vp, C.c_int64, _vp, _vp, _vp]
    L.kidmp_ppt_limbs_to_sums.restype = C.c_int
    L.kidmp_ppt_limbs_to_sums.argtypes = [C.POINTER(C.c_int64), _dp]
    L.kidmp_shard_bounds.restype = C.c_int
    L.kidmp_shard_bounds.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.kidmp_init_multi.restype = C.c_int
    L.kidmp_init_multi.argtypes = [C.POINTER(_Cfg), C.c_int32, C.POINTER(C.c_int32), C.POINTER(_vp)]
    L.kidmp_finalize_multi.restype = None
    L.kidmp_finalize_multi.argtypes = [_vp]
    L.kidmp_multi_last_error.restype = C.c_char_p
    L.kidmp_multi_last_error.argtypes = [_vp]
    L.kidmp_multi_size.restype = C.c_int32
    L.kidmp_multi_size.argtypes = [_vp]
    L.kidmp_multi_context.restype = _vp
    L.kidmp_multi_context.argtypes = [_vp, C.c_int32]
    L.kidmp_batch_step_host_multi.restype = C.c_int
    L.kidmp_batch_step_host_multi.argtypes = [_vp, C.c_int64, C.c_int32, C.c_double] + [_dp] * 17 + [C.POINTER(C.c_int32), _dp]
    L.kidmp_batch_step_host_multi_diag.restype = C.c_int
    L.kidmp_batch_step_host_multi_diag.argtypes = [_vp, C.c_int64, C.c_int32, C.c_double] + [_dp] * 17 + [C.POINTER(C.c_int32), _dp, _dp]
    L.kidmp_reserve.restype = C.c_int
    L.kidmp_reserve.argtypes = [_vp, C.c_int64, C.c_int32]
    L.kidmp_math_probe.restype = C.c_int
    L.kidmp_math_probe.argtypes = [_vp, C.c_int32, C.c_int64, _dp, _dp, _dp]
    L.kidmp_get_table.restype = C.c_int64
    L.kidmp_get_table.argtypes = [_vp, C.c_char_p, _dp, C.c_int64]
    L.kidmp_get_const.restype = C.c_int64
    L.kidmp_get_const.argtypes = [_vp, C.c_char_p, _dp, C.c_int64]
    L.kidmp_save_table_cache.restype = C.c_int
    L.kidmp_save_table_cache.argtypes = [_vp, C.c_char_p]
    L.kidmp_load_table_cache.restype = C.c_int
    L.kidmp_load_table_cache.argtypes = [_vp, C.c_char_p]
    L.kidmp_table_cache_reuse.restype = C.c_int
    L.kidmp_table_cache_reuse.argtypes = [_vp, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.kidmp_cache_write_file.restype = C.c_int
    L.kidmp_cache_write_file.argtypes = [C.c_char_p, C.c_int32, C.POINTER(_dp), C.c_int64]
    L.kidmp_cache_read_file.restype = C.c_int
    L.kidmp_cache_read_file.argtypes = [C.c_char_p, C.c_int32, C.POINTER(_dp), C.c_int64]
    L.kidmp_host_alloc.restype = _vp
    L.kidmp_host_alloc.argtypes = [C.c_size_t]
    L.kidmp_host_free.restype = None
    L.kidmp_host_free.argtypes = [_vp]
    L.kidmp_set_host_chunk.restype = C.c_int
    L.kidmp_set_host_chunk.argtypes = [_vp, C.c_int64]
    L.kidmp_init_seconds.restype = C.c_double
    L.kidmp_init_seconds.argtypes = [_vp]
    L.kidmp_kernel_name.restype = C.c_char_p
    L.kidmp_kernel_name.argtypes = []
    _lib = L
    return L


def _np_ptr(a):
    return a.ctypes.data_as(_dp)


class _PinnedBlock:
    """Owner of one kidmp_host_alloc block.  numpy arrays made from it (and their views) hold it as their base,
    so the block is freed when the last of them goes away."""

    def __init__(self, nbytes):
        self.nbytes = max(int(nbytes), 1)
        self.ptr = load_library().kidmp_host_alloc(self.nbytes)
        if not self.ptr:
            raise KidmpError("kidmp_host_alloc(%d) failed: %s" % (nbytes, load_library().kidmp_last_error(None).decode()))
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.kidmp_host_free(self.ptr)
            self.ptr = None


def host_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (kidmp_host_alloc): what the host-array entries can move by DMA."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    raw = np.asarray(_PinnedBlock(n * dtype.itemsize))       # base = the block
    return raw[: n * dtype.itemsize].view(dtype).reshape(shape)


def host_pinned_copy(a):
    """A page-locked copy of a numpy array."""
    b = host_empty(a.shape, a.dtype)
    b[...] = a
    return b


class ThompsonMP:
    """One context = the module state of module_mp_thompson09n after thompson_init:
    constants on the host, lookup tables resident in HBM."""

    def __init__(self, iiwarm=False, set_Nc=100.0, l_sediment=True, device=0, aerosol_aware=False):
        self._h = None
        L = load_library()
        cfg = _Cfg(int(bool(iiwarm)), int(bool(l_sediment)), float(set_Nc), int(device), int(bool(aerosol_aware)))
        self.aerosol_aware = bool(aerosol_aware)
        h = _vp()
        rc = L.kidmp_init(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise KidmpError("kidmp_init failed (%d): %s" % (rc, L.kidmp_last_error(None).decode()))
        self._h = h
        self.iiwarm = bool(iiwarm)
        self.device = int(device)
        self.init_seconds = L.kidmp_init_seconds(h)

    def close(self):
        if getattr(self, "_h", None):
            load_library().kidmp_finalize(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise KidmpError("kidmp error %d: %s" % (rc, load_library().kidmp_last_error(self._h).decode()))
        return rc

    # ---- mp_thompson(qv1d, ..., dt): host arrays, one column (M:1156-1177) ----
    def mp_thompson(self, qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d, nifa1d, t1d,
                    p1d, w1d, dzq, pptrain=0.0, pptsnow=0.0, pptgraul=0.0, pptice=0.0, dt=10.0):
        """Arrays (float64, length nz) are updated in place like the Fortran INOUT dummies.
        Returns the accumulated (pptrain, pptsnow, pptgraul, pptice)."""
        arrs = [qv1d, qc1d, qi1d, qr1d, qs1d, qg1d, ni1d, nr1d, nc1d, nwfa1d, nifa1d, t1d, p1d, w1d, dzq]
        nz = len(qv1d)
        for a in arrs:
            if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.shape == (nz,)):
                raise KidmpError("mp_thompson: arrays must be contiguous float64 of one length")
        ppt = np.array([pptrain, pptsnow, pptgraul, pptice], dtype=np.float64)
        self._check(load_library().kidmp_column_step(self._h, nz, float(dt), *[_np_ptr(a) for a in arrs], _np_ptr(ppt)))
        return tuple(ppt)

    # ---- batched host entry: numpy [ncol, nz] ----
    def batch_step_host(self, st, dt, ppt=None, want_rates=False):
        """numpy float64 [ncol, nz] arrays, in place.  Keys KiD itself never fills may be missing (or None): nc, nwfa,
        nifa and w for a context without aerosol_aware, qi, qs, qg, ni for an iiwarm context (include/kidmp.h)."""
        ncol, nz = st["qv"].shape
        ptrs = []
        for k in STATE_NAMES + FORCING_NAMES:
            a = st.get(k)
            if a is None:
                ptrs.append(None)
                continue
            if not (a.dtype == np.float64 and a.flags.c_contiguous and a.shape == (ncol, nz)):
                raise KidmpError("batch_step_host: %s must be contiguous float64 [ncol, nz]" % k)
            ptrs.append(_np_ptr(a))
        if ppt is None:
            ppt = np.zeros((ncol, 4))
        rates = np.zeros((ncol, NRATES, nz)) if want_rates else None
        self._check(load_library().kidmp_batch_step_host(
            self._h, ncol, nz, float(dt), *ptrs, _np_ptr(ppt), _np_ptr(rates) if want_rates else None))
        return ppt, rates

    def _want(self, a, dtype, shape, what):
        """A device-entry argument: contiguous CUDA tensor of the given dtype/shape on THIS context's GPU."""
        if not (a.is_cuda and a.dtype == dtype and a.is_contiguous() and tuple(a.shape) == tuple(shape)):
            raise KidmpError("%s must be a contiguous %s CUDA tensor %s" % (what, str(dtype).replace("torch.", ""), list(shape)))
        if a.device.index != self.device:
            raise KidmpError("%s lives on cuda:%d but this context is bound to cuda:%d" % (what, a.device.index, self.device))

    # ---- batched device entry: torch CUDA tensors [ncol, nz], in place ----
    def batch_step(self, st, dt, ppt, rates=None, nstep=None, stream=None):
        """st: dict of float64 CUDA tensors [ncol, nz] (STATE_NAMES + p, dz; w optional).
        ppt: float64 [ncol, 4], accumulated in place.  Asynchronous on `stream`
        (default: torch's current stream)."""
        import torch
        q = st["qv"]
        ncol, nz = q.shape
        w = st.get("w")
        for k in STATE_NAMES + ("p", "dz"):
            self._want(st[k], torch.float64, (ncol, nz), "batch_step: " + k)
        if w is not None:
            self._want(w, torch.float64, (ncol, nz), "batch_step: w")
        self._want(ppt, torch.float64, (ncol, 4), "batch_step: ppt")
        if rates is not None:
            self._want(rates, torch.float64, (ncol, NRATES, nz), "batch_step: rates")
        if nstep is not None:
            self._want(nstep, torch.int32, (ncol, 4), "batch_step: nstep")
        s = stream if stream is not None else torch.cuda.current_stream(q.device).cuda_stream
        args = [st[k].data_ptr() for k in STATE_NAMES] + [st["p"].data_ptr(), w.data_ptr() if w is not None else None,
                                                          st["dz"].data_ptr(), ppt.data_ptr(),
                                                          rates.data_ptr() if rates is not None else None,
                                                          nstep.data_ptr() if nstep is not None else None]
        self._check(load_library().kidmp_batch_step_device(self._h, ncol, nz, float(dt), *args, s))

    # ---- binary32 state: the reference's native arithmetic ("p32n": REAL = binary32, DOUBLE PRECISION = binary64)
    #      and the all-binary32 build ("f32") -- include/kidmp.h, kidmp32_* ----
    ARITH = {"p32n": 0, "f32": 1}

    def batch_step32_host(self, st, dt, arith="p32n", ppt=None, want_rates=False, want_nstep=False):
        """numpy float32 [ncol, nz] arrays, in place.  Returns (ppt float32 [ncol, 4], rates float64 or None, nstep or None)."""
        ncol, nz = st["qv"].shape
        fpp = C.POINTER(C.c_float)
        for k in STATE_NAMES + FORCING_NAMES:
            a = st[k]
            if not (a.dtype == np.float32 and a.flags.c_contiguous and a.shape == (ncol, nz)):
                raise KidmpError("batch_step32_host: %s must be contiguous float32 [ncol, nz]" % k)
        if ppt is None:
            ppt = np.zeros((ncol, 4), dtype=np.float32)
        rates = np.zeros((ncol, NRATES, nz)) if want_rates else None
        nstep = np.zeros((ncol, 4), dtype=np.int32) if want_nstep else None
        self._check(load_library().kidmp32_batch_step_host(
            self._h, ncol, nz, float(dt), *[st[k].ctypes.data_as(fpp) for k in STATE_NAMES + FORCING_NAMES],
            ppt.ctypes.data_as(fpp), _np_ptr(rates) if want_rates else None,
            nstep.ctypes.data_as(C.POINTER(C.c_int32)) if want_nstep else None, self.ARITH[arith]))
        return ppt, rates, nstep

    def batch_step32(self, st, dt, ppt, arith="p32n", rates=None, nstep=None, stream=None):
        """torch float32 CUDA tensors [ncol, nz], in place, asynchronous (the device entry of the binary32 builds)."""
        import torch
        q = st["qv"]
        ncol, nz = q.shape
        for k in STATE_NAMES + ("p", "dz"):
            self._want(st[k], torch.float32, (ncol, nz), "batch_step32: " + k)
        self._want(ppt, torch.float32, (ncol, 4), "batch_step32: ppt")
        if rates is not None:
            self._want(rates, torch.float64, (ncol, NRATES, nz), "batch_step32: rates")
        if nstep is not None:
            self._want(nstep, torch.int32, (ncol, 4), "batch_step32: nstep")
        s = stream if stream is not None else torch.cuda.current_stream(q.device).cuda_stream
        w = st.get("w")
        args = [st[k].data_ptr() for k in STATE_NAMES] + [st["p"].data_ptr(), w.data_ptr() if w is not None else None,
                                                          st["dz"].data_ptr(), ppt.data_ptr(),
                                                          rates.data_ptr() if rates is not None else None,
                                                          nstep.data_ptr() if nstep is not None else None]
        self._check(load_library().kidmp32_batch_step_device(self._h, ncol, nz, float(dt), *args, self.ARITH[arith], s))

    def default_aerosols(self, qv, t, p, stream=None):
        """nc, nwfa, nifa for the inputs the KiD wrapper leaves unset (W:36; formulas M:958-964)."""
        import torch
        for name, a in (("qv", qv), ("t", t), ("p", p)):
            self._want(a, torch.float64, tuple(qv.shape), "default_aerosols: " + name)
        nc, nwfa, nifa = torch.empty_like(qv), torch.empty_like(qv), torch.empty_like(qv)
        s = stream if stream is not None else torch.cuda.current_stream(qv.device).cuda_stream
        self._check(load_library().kidmp_default_aerosols_device(
            self._h, qv.numel(), qv.data_ptr(), t.data_ptr(), p.data_ptr(), nc.data_ptr(), nwfa.data_ptr(),
            nifa.data_ptr(), s))
        return nc, nwfa, nifa

    def set_host_chunk(self, ncol_per_chunk):
        """Columns per pipeline chunk of the host-array entries (0 = default)."""
        self._check(load_library().kidmp_set_host_chunk(self._h, int(ncol_per_chunk)))

    def reduce_ppt(self, ppt, stream=None):
        """Domain sums of the surface precipitation on the device (W:248-275 analogue)."""
        import torch
        self._want(ppt, torch.float64, (ppt.shape[0], 4), "reduce_ppt: ppt")
        out = torch.empty(4, dtype=torch.float64, device=ppt.device)
        s = stream if stream is not None else torch.cuda.current_stream(ppt.device).cuda_stream
        self._check(load_library().kidmp_reduce_ppt_device(self._h, ppt.shape[0], ppt.data_ptr(), out.data_ptr(), s))
        return out

    def reduce_ppt_exact(self, ppt, stream=None):
        """The same four sums as exact fixed-point accumulators: int64 [PPT_LIMBS] on the device.  Integer sums are
        associative: multi-GPU callers all-reduce(SUM) the limbs and get identical bits for every partition of the
        columns; limbs_to_sums() converts."""
        import torch
        self._want(ppt, torch.float64, (ppt.shape[0], 4), "reduce_ppt_exact: ppt")
        out = torch.empty(PPT_LIMBS, dtype=torch.int64, device=ppt.device)
        s = stream if stream is not None else torch.cuda.current_stream(ppt.device).cuda_stream
        self._check(load_library().kidmp_reduce_ppt_exact_device(self._h, ppt.shape[0], ppt.data_ptr(), out.data_ptr(), s))
        return out

    def reduce_rates(self, rates, stream=None):
        """Sum over columns of the rate diagnostics: [ncol, 36, nz] -> [36, nz] (the nx-mean profiles KiD plots are
        this / ncol).  Fixed summation order."""
        import torch
        ncol, nr, nz = rates.shape
        self._want(rates, torch.float64, (ncol, NRATES, nz), "reduce_rates: rates")
        out = torch.empty(NRATES, nz, dtype=torch.float64, device=rates.device)
        s = stream if stream is not None else torch.cuda.current_stream(rates.device).cuda_stream
        self._check(load_library().kidmp_reduce_rates_device(self._h, ncol, nz, rates.data_ptr(), out.data_ptr(), s))
        return out

    def reserve(self, ncol, nz=120):
        """Deprecated no-op (kidmp_reserve): the step owns no per-batch device memory.  Kept so that hosts written
        against rounds 1-2 keep working; validates its arguments like every entry."""
        self._check(load_library().kidmp_reserve(self._h, int(ncol), int(nz)))

    SANITY_MAX = ("qc", "qr", "nr", "qs", "qi", "qg", "ni")
    SANITY_NEG = ("qc", "qr", "nr", "qs", "qi", "qg", "ni", "qv")

    def sanity(self, st, stream=None):
        """The post-step scan of the scheme's 3-D driver (M:1025-1094): [15] float64 on the device =
        maxima of SANITY_MAX, then counts of negative entries of SANITY_NEG."""
        import torch
        q = st["qv"]
        for k in self.SANITY_NEG:
            self._want(st[k], torch.float64, tuple(q.shape), "sanity: " + k)
        out = torch.empty(15, dtype=torch.float64, device=q.device)
        s = stream if stream is not None else torch.cuda.current_stream(q.device).cuda_stream
        self._check(load_library().kidmp_sanity_device(self._h, q.numel(), *[st[k].data_ptr() for k in self.SANITY_NEG],
                                                       out.data_ptr(), s))
        return out

    def effective_radii(self, st, preset=(2.49e-6, 4.99e-6, 9.99e-6), stream=None):
        """calc_effectRad (M:4834-4935): (re_qc, re_qi, re_qs) [ncol, nz] on the device, started from the presets of the
        scheme's driver (M:1111-1113)."""
        import torch
        q = st["qv"]
        for k in ("t", "p", "qv", "qc", "nc", "qi", "ni", "qs"):
            self._want(st[k], torch.float64, tuple(q.shape), "effective_radii: " + k)
        out = [torch.full_like(q, v) for v in preset]
        s = stream if stream is not None else torch.cuda.current_stream(q.device).cuda_stream
        self._check(load_library().kidmp_effective_radii_device(
            self._h, q.numel(), *[st[k].data_ptr() for k in ("t", "p", "qv", "qc", "nc", "qi", "ni", "qs")],
            *[o.data_ptr() for o in out], s))
        return tuple(out)

    def kernel_fingerprint(self, arith="p64"):
        """'src:<hash>;vgpr:<n>;lds:<bytes>;scratch:<bytes>' of this context's nz <= 120 column-step kernel, in the
        parity arithmetic (p64) or one of the binary32 ones (p32n, f32)."""
        if arith == "p64":
            return load_library().kidmp_kernel_fingerprint(self._h).decode()
        return load_library().kidmp32_kernel_fingerprint(self._h, {"p32n": 0, "f32": 1}[arith]).decode()

    # ---- introspection for parity tests ----
    MATH_FUNCS = ("log", "log10", "exp", "exp10", "sqrt", "cbrt", "pow", "rcp_seed", "div", "ieee_div", "rcp")

    def math_probe(self, fn, x, y=None):
        """Evaluate one of the column kernel's fp64 math helpers (csrc/fastmath.h) on the device, elementwise."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(x if y is None else y, dtype=np.float64)
        out = np.empty_like(x)
        self._check(load_library().kidmp_math_probe(self._h, self.MATH_FUNCS.index(fn), x.size, _np_ptr(x), _np_ptr(y),
                                                    _np_ptr(out)))
        return out

    def table(self, name, shape=None):
        L = load_library()
        n = self._check(L.kidmp_get_table(self._h, name.encode(), None, 0))
        out = np.empty(n)
        self._check(L.kidmp_get_table(self._h, name.encode(), _np_ptr(out), n))
        return out.reshape(shape, order="F") if shape else out

    def const(self, name):
        L = load_library()
        n = self._check(L.kidmp_get_const(self._h, name.encode(), None, 0))
        out = np.empty(n)
        self._check(L.kidmp_get_const(self._h, name.encode(), _np_ptr(out), n))
        return out

    # ---- the reference's run_data/*.data table caches (M:3717-3829, M:3864-4078) ----
    def save_table_cache(self, directory):
        self._check(load_library().kidmp_save_table_cache(self._h, os.fsencode(directory)))

    def load_table_cache(self, directory):
        self._check(load_library().kidmp_load_table_cache(self._h, os.fsencode(directory)))

    def table_cache_reuse(self, directory, l_reuse, write_if_built=True):
        """thompson_init's use of run_data/*.data (M:3717-3729, M:3864-3895): read a file that exists when l_reuse,
        else write the GPU-built tables.  Returns the status bits (1, 2: racg, racs read; 4, 8: written)."""
        st = C.c_int32(0)
        self._check(load_library().kidmp_table_cache_reuse(self._h, os.fsencode(directory), int(bool(l_reuse)),
                                                           int(bool(write_if_built)), C.byref(st)))
        return st.value

    @staticmethod
    def kernel_name():
        return load_library().kidmp_kernel_name().decode()


def limbs_to_sums(limbs):
    """int64 [PPT_LIMBS] (host: numpy array or CPU tensor) -> the four precipitation domain sums (float64 numpy)."""
    a = np.ascontiguousarray(np.asarray(limbs, dtype=np.int64))
    if a.shape != (PPT_LIMBS,):
        raise KidmpError("limbs_to_sums: expected %d int64 limbs" % PPT_LIMBS)
    out = np.empty(4)
    load_library().kidmp_ppt_limbs_to_sums(a.ctypes.data_as(C.POINTER(C.c_int64)), _np_ptr(out))
    return out


def shard_bounds(ncol, nshard, shard):
    """The library's contiguous column ranges (kidmp_shard_bounds; no GPU needed)."""
    lo, hi = C.c_int64(), C.c_int64()
    rc = load_library().kidmp_shard_bounds(int(ncol), int(nshard), int(shard), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise KidmpError("kidmp_shard_bounds failed (%d): %s" % (rc, load_library().kidmp_last_error(None).decode()))
    return lo.value, hi.value


class ThompsonMulti:
    """Several GPUs behind one host-array call (kidmp_init_multi / kidmp_batch_step_host_multi): contiguous column
    ranges over the device list, one pipeline per device on its own host thread, the precipitation domain sums
    all-reduced with RCCL inside the library.  This is what the Fortran drop-in uses when more than one device is
    configured; the Python mirror exists for the tests."""

    def __init__(self, devices, iiwarm=False, set_Nc=100.0, l_sediment=True, aerosol_aware=False):
        self._h = None
        L = load_library()
        cfg = _Cfg(int(bool(iiwarm)), int(bool(l_sediment)), float(set_Nc), 0, int(bool(aerosol_aware)))
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = _vp()
        rc = L.kidmp_init_multi(C.byref(cfg), len(devices), devs, C.byref(h))
        if rc != 0:
            raise KidmpError("kidmp_init_multi failed (%d): %s" % (rc, L.kidmp_last_error(None).decode()))
        self._h = h
        self.devices = [int(d) for d in devices]
        self.iiwarm = bool(iiwarm)

    def close(self):
        if getattr(self, "_h", None):
            load_library().kidmp_finalize_multi(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def batch_step_host(self, st, dt, ppt=None, want_rates=False, want_nstep=False, want_sanity=False):
        """numpy float64 [ncol, nz] arrays, in place (optional keys as ThompsonMP.batch_step_host).
        Returns (ppt, rates or None, nstep or None, precip_sums[4]) -- with want_sanity a fifth element, the 15-number
        sanity scan of the end state (kidmp_batch_step_host_multi_diag), reduced over the devices like the sums."""
        ncol, nz = st["qv"].shape
        ptrs = []
        for k in STATE_NAMES + FORCING_NAMES:
            a = st.get(k)
            if a is None:
                ptrs.append(None)
                continue
            if not (a.dtype == np.float64 and a.flags.c_contiguous and a.shape == (ncol, nz)):
                raise KidmpError("batch_step_host: %s must be contiguous float64 [ncol, nz]" % k)
            ptrs.append(_np_ptr(a))
        if ppt is None:
            ppt = np.zeros((ncol, 4))
        rates = np.zeros((ncol, NRATES, nz)) if want_rates else None
        nstep = np.zeros((ncol, 4), dtype=np.int32) if want_nstep else None
        sums = np.zeros(4)
        L = load_library()
        a_rates = _np_ptr(rates) if want_rates else None
        a_nstep = nstep.ctypes.data_as(C.POINTER(C.c_int32)) if want_nstep else None
        if want_sanity:
            sanity = np.zeros(15)
            rc = L.kidmp_batch_step_host_multi_diag(self._h, ncol, nz, float(dt), *ptrs, _np_ptr(ppt), a_rates, a_nstep,
                                                    _np_ptr(sums), _np_ptr(sanity))
        else:
            rc = L.kidmp_batch_step_host_multi(self._h, ncol, nz, float(dt), *ptrs, _np_ptr(ppt), a_rates, a_nstep, _np_ptr(sums))
        if rc != 0:
            raise KidmpError("kidmp_batch_step_host_multi failed (%d): %s" % (rc, L.kidmp_multi_last_error(self._h).decode()))
        return (ppt, rates, nstep, sums, sanity) if want_sanity else (ppt, rates, nstep, sums)


# ---- module-level mirror of the Fortran module procedures ----
_module_ctx = None


def thompson_init(iiwarm=False, set_Nc=100.0, l_sediment=True, device=0):
    """thompson_init (M:374): builds the module-level context once, like the
    `micro_unset` guard of the KiD adapter (W:100-103)."""
    global _module_ctx
    if _module_ctx is None:
        _module_ctx = ThompsonMP(iiwarm=iiwarm, set_Nc=set_Nc, l_sediment=l_sediment, device=device)
    return _module_ctx


def mp_thompson(*args, **kw):
    """mp_thompson (M:1156) on the module-level context."""
    if _module_ctx is None:
        raise KidmpError("mp_thompson called before thompson_init")
    return _module_ctx.mp_thompson(*args, **kw)


def cache_write_file(path, tables):
    """Write `tables` (list of equally sized float64 arrays, Fortran element order) in the list-directed
    text format of the reference's `write(12,*) table` statements (M:3823-3828)."""
    arrs = [np.ascontiguousarray(np.asarray(t, dtype=np.float64).ravel(order="F")) for t in tables]
    n = arrs[0].size
    if any(a.size != n for a in arrs):
        raise KidmpError("cache_write_file: tables must have equal sizes")
    ptrs = (_dp * len(arrs))(*[_np_ptr(a) for a in arrs])
    rc = load_library().kidmp_cache_write_file(os.fsencode(path), len(arrs), ptrs, n)
    if rc != 0:
        raise KidmpError("cache_write_file failed (%d): %s" % (rc, load_library().kidmp_last_error(None).decode()))


def cache_read_file(path, ntab, n_each):
    """Read ntab tables of n_each values written by a Fortran `write(u,*)` (or by cache_write_file)."""
    arrs = [np.empty(n_each) for _ in range(ntab)]
    ptrs = (_dp * ntab)(*[_np_ptr(a) for a in arrs])
    rc = load_library().kidmp_cache_read_file(os.fsencode(path), ntab, ptrs, n_each)
    if rc != 0:
        raise KidmpError("cache_read_file failed (%d): %s" % (rc, load_library().kidmp_last_error(None).decode()))
    return arrs
