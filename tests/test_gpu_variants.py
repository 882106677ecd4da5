"""Parity across the kernel's compile-time variants and run-time switches (-m gpu): other nz (NJ = 1..4 level
groups, every LDS stride), other dt (incl. the dt > 120 s re-routing of M:2277), l_sediment off, other set_Nc,
degenerate batch sizes, and the C ABI's argument checks."""
import ctypes as C

import numpy as np
import pytest

import cases
import kat_cases as kc
from parity import FLOORS, OUT, TOL, assert_parity, max_rel

pytestmark = pytest.mark.gpu


def _resample(col, nz):
    """KAT-A style column on nz levels spanning the same 15 km (linear interpolation in level index)."""
    x0 = np.linspace(0.0, 1.0, col["qv"].shape[0])
    x1 = np.linspace(0.0, 1.0, nz)
    out = {k: np.interp(x1, x0, v) for k, v in col.items()}
    out["dz"] = np.full(nz, 15000.0 / nz)
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def _batch(cols):
    return {k: np.ascontiguousarray(np.stack([c[k] for c in cols])) for k in cases.KEYS}


def _compare(m, o, st, dt, tol=TOL, max_excluded_frac=0.08, depletion_aware=False):
    got = {k: v.copy() for k, v in st.items()}
    gppt, _ = m.batch_step_host(got, dt)
    # With long steps a species can be depleted to 1e-7 of its input in one call (q + qten*dt); the remainder
    # then carries the input's rounding error, amplified by that ratio, and the number of a depleted species is
    # rebuilt from that remainder (M:3655-3664).  depletion_aware measures such values against 1e-5 of the input.
    # Every level is checked; levels on the reference's residue-decided tests against both outcomes (parity.py).
    assert_parity(o, st, dt, got, gppt, tol=tol, depletion=1e-5 if depletion_aware else 0.0,
                  max_branch_frac=max_excluded_frac)


@pytest.mark.parametrize("nz", [2, 17, 40, 64, 65, 100, 119, 121, 128, 129, 150, 192, 200, 256])
def test_other_level_counts(gpu_mixed, oracle_mixed, nz):
    cols = [_resample(kc.kat_a(True), nz), _resample(kc.kat_a(False), nz), _resample(kc.kat_c(), nz)]
    _compare(gpu_mixed, oracle_mixed, _batch(cols), 10.0)


@pytest.mark.parametrize("nz", [2, 33, 64, 65, 120, 121, 128, 190, 256])
def test_other_level_counts_warm(gpu_warm, oracle_warm, nz):
    """The warm-rain instantiation (13-slot LDS layout) at every level-group count / LDS stride, with and without
    frozen species in the input (it reads them again from memory where the mixed-phase kernel has them in LDS)."""
    cols = [_resample(kc.kat_a(False), nz), _resample(kc.kat_a(True), nz), _resample(kc.kat_c(), nz),
            _resample(kc.kat_a(False), nz), _resample(kc.kat_a(False), nz)]
    cols[3]["qr"] = cols[3]["qr"] * 4.0
    _compare(gpu_warm, oracle_warm, _batch(cols), 10.0)


@pytest.mark.parametrize("dt", [1.0, 5.0, 30.0, 60.0, 150.0])
def test_other_time_steps(gpu_mixed, oracle_mixed, dt):
    st = cases.edge_cases()
    _compare(gpu_mixed, oracle_mixed, st, dt, max_excluded_frac=0.2, depletion_aware=dt > 10.0)


def test_switches_l_sediment_off_and_other_set_nc():
    from kid_amd import ThompsonMP
    from oracle.oracle import Oracle
    for kw in (dict(iiwarm=False, set_Nc=100.0, l_sediment=False), dict(iiwarm=False, set_Nc=300.0, l_sediment=True),
               dict(iiwarm=True, set_Nc=50.0, l_sediment=True)):
        m, o = ThompsonMP(**kw), Oracle(**kw)
        try:
            st = cases.edge_cases()
            rho = 0.622 * st["p"] / (287.04 * st["t"] * (st["qv"] + 0.622))
            st["nc"] = kw["set_Nc"] * 1e6 / rho
            if kw["iiwarm"]:
                for k in ("qi", "ni", "qs", "qg"):
                    st[k][:] = 0.0
            _compare(m, o, st, 10.0, max_excluded_frac=0.2)
        finally:
            m.close()
            o.close()


def test_degenerate_batches(gpu_mixed, oracle_mixed):
    one = {k: v[:1].copy() for k, v in cases.edge_cases().items()}
    _compare(gpu_mixed, oracle_mixed, one, 10.0)
    empty = {k: np.zeros((0, 120)) for k in cases.KEYS}
    ppt, _ = gpu_mixed.batch_step_host(empty, 10.0)
    assert ppt.shape == (0, 4)


def test_c_abi_argument_checks(gpu_mixed):
    from kid_amd import KidmpError
    from kid_amd.thompson import load_library
    L = load_library()
    st = {k: v[:1].copy() for k, v in cases.edge_cases().items()}
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step_host({k: v[:, :1].copy() for k, v in st.items()}, 10.0)       # nz = 1
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step_host({k: np.zeros((1, 300)) for k in cases.KEYS}, 10.0)       # nz > KIDMP_MAX_NZ
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step_host(st, 0.0)                                                 # dt <= 0
    dp = C.POINTER(C.c_double)
    args = [st[k].ctypes.data_as(dp) for k in cases.KEYS]
    args[3] = None                                                                          # null qr pointer
    ppt = np.zeros(4)
    rc = L.kidmp_batch_step_host(gpu_mixed._h, 1, 120, 10.0, *args, ppt.ctypes.data_as(dp), None)
    assert rc == -1 and b"null" in L.kidmp_last_error(gpu_mixed._h)
    assert L.kidmp_batch_step_host(None, 1, 120, 10.0, *args, ppt.ctypes.data_as(dp), None) == -5


@pytest.mark.parametrize("ncol", [1, 2, 3, 5, 6, 7, 9, 13])
def test_workgroup_remainders_and_dry_columns(gpu_mixed, oracle_mixed, ncol):
    """A workgroup holds 4 columns that share the rate sweep: every remainder of ncol mod 4, with a column without
    microphysics (early return of M:1540) at each position of the workgroup in turn."""
    ec = cases.edge_cases()
    n_ec = ec["qv"].shape[0]
    dry = {k: v[:1].copy() for k, v in ec.items()}
    for k in ("qc", "qi", "qr", "qs", "qg", "ni", "nr"):
        dry[k][:] = 0.0
    dry["qv"][:] = 1.0e-6                                            # far below saturation everywhere
    for pos in range(min(ncol, 4)):
        cols = [{k: v[(i * 5 + pos) % n_ec] for k, v in ec.items()} for i in range(ncol)]
        cols[pos] = {k: v[0] for k, v in dry.items()}
        st = {k: np.ascontiguousarray(np.stack([c[k] for c in cols])) for k in cases.KEYS}
        _compare(gpu_mixed, oracle_mixed, st, 10.0)


def test_hip_graph_capture_of_steps(gpu_mixed):
    """The device entry never allocates (no per-batch work buffer), so K steps of a batch size the context has never
    seen can be captured into a HIP graph and replayed; the replay gives the same bits as K eager launches."""
    import torch
    st = cases.config3(72)                                   # no other test steps 72 columns
    graphed = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    ppt_g = torch.zeros(72, 4, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):                                # the very first launches of this size are the captured ones
        for _ in range(3):
            gpu_mixed.batch_step(graphed, 10.0, ppt_g)
    for k in graphed:                                                  # capture does not execute: state still initial
        assert torch.equal(graphed[k].cpu(), torch.from_numpy(st[k]))
    g.replay()
    torch.cuda.synchronize()
    eager = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    ppt_e = torch.zeros(72, 4, dtype=torch.float64, device="cuda")
    for _ in range(3):
        gpu_mixed.batch_step(eager, 10.0, ppt_e)
    torch.cuda.synchronize()
    for k in cases.KEYS:
        assert torch.equal(graphed[k], eager[k]), k
    assert torch.equal(ppt_g, ppt_e)


def test_one_context_steps_on_two_streams_at_once(gpu_mixed):
    """No context-owned work buffer: two batches stepped concurrently on two streams through ONE context end exactly
    where they end when stepped one after the other."""
    import torch
    a0, b0 = cases.config3(3000, seed=cases.SEED + 21), cases.config5(3000, seed=cases.SEED + 22)

    def run(concurrent):
        A = {k: torch.from_numpy(v).cuda() for k, v in a0.items()}
        B = {k: torch.from_numpy(v).cuda() for k, v in b0.items()}
        pa = torch.zeros(3000, 4, dtype=torch.float64, device="cuda")
        pb = torch.zeros(3000, 4, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        s1, s2 = (torch.cuda.Stream(), torch.cuda.Stream()) if concurrent else (torch.cuda.current_stream(),) * 2
        for _ in range(4):
            with torch.cuda.stream(s1):
                gpu_mixed.batch_step(A, 10.0, pa)
            with torch.cuda.stream(s2):
                gpu_mixed.batch_step(B, 10.0, pb)
        torch.cuda.synchronize()
        return A, B, pa, pb

    A1, B1, pa1, pb1 = run(False)
    A2, B2, pa2, pb2 = run(True)
    for k in cases.KEYS:
        assert torch.equal(A1[k], A2[k]) and torch.equal(B1[k], B2[k]), k
    assert torch.equal(pa1, pa2) and torch.equal(pb1, pb2)


def test_c_abi_argument_checks_binary32_and_diagnostics(gpu_mixed):
    """The round-2 entries refuse bad arguments with a status code, like the rest of the C ABI."""
    import torch
    from kid_amd import KidmpError
    from kid_amd.thompson import load_library
    L = load_library()
    st = {k: np.ascontiguousarray(v[:2].astype(np.float32)) for k, v in cases.edge_cases().items()}
    fp = C.POINTER(C.c_float)
    args = [st[k].ctypes.data_as(fp) for k in cases.KEYS]
    ppt = np.zeros((2, 4), dtype=np.float32)
    rc = L.kidmp32_batch_step_host(gpu_mixed._h, 2, 120, 10.0, *args, ppt.ctypes.data_as(fp), None, None, 7)      # unknown arith
    assert rc == -1 and b"arith" in L.kidmp_last_error(gpu_mixed._h)
    rc = L.kidmp32_batch_step_host(gpu_mixed._h, 2, 120, -1.0, *args, ppt.ctypes.data_as(fp), None, None, 0)      # dt <= 0
    assert rc == -1
    assert L.kidmp32_batch_step_host(None, 2, 120, 10.0, *args, ppt.ctypes.data_as(fp), None, None, 0) == -5      # no context
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step32(dev, 10.0, torch.zeros(2, 4, dtype=torch.float64, device="cuda"))                   # ppt must be float32
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step(dev, 10.0, torch.zeros(2, 4, dtype=torch.float64, device="cuda"))                     # float32 state into the fp64 entry
    assert L.kidmp_reduce_rates_device(gpu_mixed._h, 4, 1, None, None, None) == -1
    assert L.kidmp_sanity_device(gpu_mixed._h, 10, *([None] * 9), None) == -1
    assert L.kidmp_effective_radii_device(gpu_mixed._h, 10, *([None] * 11), None) == -1
    # an empty batch is fine everywhere
    e32 = {k: torch.zeros(0, 120, dtype=torch.float32, device="cuda") for k in cases.KEYS}
    gpu_mixed.batch_step32(e32, 10.0, torch.zeros(0, 4, dtype=torch.float32, device="cuda"))
