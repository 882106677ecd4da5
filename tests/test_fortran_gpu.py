"""BASELINE config 1 through the Fortran boundary (-m gpu): the mini KiD driver calls
mphys_thompson09_interfacen -> mp_thompson_batch -> ISO_C_BINDING -> libkidmp.so -> HIP."""
import os
import subprocess

import numpy as np
import pytest

import kat_cases as kc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "kid_amd", "fortran", "build", "kid_mini_driver")


def _run(nx, nsteps=360):
    assert os.path.exists(EXE), "build the Fortran shim first (__graft_entry__.build())"
    out = subprocess.run([EXE, str(nx), str(nsteps)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    vals = {}
    for line in out.stdout.splitlines():
        p = line.split()
        if p and p[0] in ("KATB", "KATBN"):
            vals[p[0]] = np.array([float(x) for x in p[1:5]])
    return vals


def test_kid_warm_case_360_steps_matches_reference_kat_and_oracle(oracle_warm):
    got = _run(1)["KATB"]
    # the reference's own end state for this case (SURVEY 9h KAT-B, 7 digits)
    for g, r in zip(got, (1.530434, 2.218719e-2, 2.694135e-3, 1.060568e6)):
        assert abs(g / r - 1) < 1e-6
    # and the oracle's adapter, step for step
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
    ref = np.array([qv.sum(), hy[0, 0].sum(), hy[0, 1].sum(), hy[1, 1].sum()])
    np.testing.assert_allclose(got, ref, rtol=1e-9)


def test_batched_adapter_columns_are_independent():
    v = _run(5, 60)
    assert np.array_equal(v["KATB"], v["KATBN"])        # replicated columns stay bit-identical


def test_large_batches_through_the_fortran_drop_in():
    """nx = 3000 columns (a 43 MB state: the adapter's work arrays must not live on the stack, the host-array call
    runs as several pipeline chunks) end where nx = 5 ends, column for column."""
    small, big = _run(5, 6), _run(3000, 6)
    assert np.array_equal(big["KATB"], big["KATBN"])
    assert np.array_equal(big["KATB"], small["KATB"])


def test_default_real4_kid_build_runs_through_the_same_shim():
    """KiD's native build has 4-byte default REAL (the reference's "P32n" arithmetic keeps its state in fp32).
    The shim converts at the boundary, so the same modules serve that build; the end state then differs from the
    fp64 run only by the fp32 storage of the state between calls (8e-5 on sum(qc) for this case)."""
    exe32 = os.path.join(ROOT, "kid_amd", "fortran", "build32", "kid_mini_driver")
    if not os.path.exists(exe32):
        pytest.skip("build32 not built (make -C kid_amd/fortran FFLAGS=-O2 B=build32)")
    out = subprocess.run([exe32, "1", "360"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    got = None
    for line in out.stdout.splitlines():
        p = line.split()
        if p and p[0] == "KATB":
            got = np.array([float(x) for x in p[1:5]])
    ref = np.array([1.530434, 2.218719e-2, 2.694135e-3, 1.060568e6])       # reference P64 end state
    native = np.array([1.530434, 2.218541e-2, 2.693803e-3, 1.060634e6])    # reference native P32n end state
    assert got is not None
    assert np.all(np.abs(got / ref - 1) < 3e-4) and np.all(np.abs(got / native - 1) < 3e-4), got


# ---------------------------------------------------------------------------------------------------------------
# save_dg side effects of the drop-in (SURVEY 8b "side effects", 8f1): the 36 process rates of M:2962-3124 and the
# surface-precipitation calls of W:155-182 / W:248-303, recorded by the stand-in `diagnostics` module
from kid_amd import RATE_NAMES  # noqa: E402


def _run_dump(tmp_path, nx, nsteps, case, dump_step):
    out = subprocess.run([EXE, str(nx), str(nsteps), case, str(dump_step)], capture_output=True, text=True, timeout=600,
                         cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    log = []
    for line in open(os.path.join(str(tmp_path), "dg_dump.txt")):
        p = line.split()
        # form name k i value units... dim   (units may contain a blank: 'kg/kg m')
        log.append(dict(form=p[0], name=p[1], k=int(p[2]), i=int(p[3]), v=float(p[4]), units=" ".join(p[5:-1]), dim=p[-1]))
    inp = np.loadtxt(os.path.join(str(tmp_path), "dg_inputs.txt")).reshape(nx, 120, 12)
    return log, inp


def _oracle_rates_for(oracle, inp, iiwarm):
    """What the adapter feeds mp_thompson (W:59-97 + the U2 defaults), then one oracle step with its rate buffer."""
    theta, exner, qv, dz = inp[:, 0], inp[:, 1], inp[:, 2], inp[:, 3]
    st = dict(t=theta * exner, p=1.0e5 * exner ** (1.0 / (287.058 / 1005.0)), qv=qv.copy(), dz=dz.copy(), w=np.zeros(120),
              qc=inp[:, 4].copy(), qr=inp[:, 5].copy(), nr=inp[:, 6].copy())
    for j, k in enumerate(("qi", "ni", "qs", "qg")):
        st[k] = np.zeros(120) if iiwarm else inp[:, 7 + j].copy()
    rho = 0.622 * st["p"] / (287.04 * st["t"] * (st["qv"] + 0.622))
    st["nc"], st["nwfa"], st["nifa"] = 100.0e6 / rho, 11.1e6 / rho, 0.5e6 * 0.01 / rho
    st = {k: np.ascontiguousarray(v) for k, v in st.items()}
    ppt, rates, _, no_micro = oracle.column_step(st, 10.0, want_rates=True)
    return ppt, rates, no_micro


def _check_rate_values(got, ref, what):
    scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-9 * scale)
    # inputs reach the two sides through different pow/exner evaluations (Fortran runtime vs numpy): 1e-8
    assert np.max(err) < 1e-8, (what, float(np.max(err)), RATE_NAMES[int(np.argmax(np.max(err, axis=1)))])


def test_rate_diagnostics_warm_nx1_names_order_values(tmp_path, oracle_warm):
    """nx == 1, iiwarm: per level the six warm rates in the order of M:3104-3119 through save_dg(k, value, ...),
    then the five scalar precipitation calls of W:155-182."""
    log, inp = _run_dump(tmp_path, 1, 95, "warm", 90)
    rates = [e for e in log if e["form"] == "k"]
    assert len(rates) == 6 * 120 and log[: 6 * 120] == rates                     # emitted before the precipitation calls
    want = RATE_NAMES[30:]
    for n, e in enumerate(rates):
        assert e["name"] == want[n % 6] and e["k"] == n // 6 + 1 and e["units"] == "/kg/s" and e["dim"] == "z", (n, e)
    rppt, rref, no_micro = _oracle_rates_for(oracle_warm, inp[0], True)
    assert not no_micro
    got = np.zeros((36, 120))
    for e in rates:
        got[RATE_NAMES.index(e["name"]), e["k"] - 1] = e["v"]
    assert np.count_nonzero(got[30:]) > 100
    _check_rate_values(got[30:], rref[30:], "warm")
    tail = log[6 * 120:]
    assert [e["name"] for e in tail] == ["surface_ppt_for_rain", "surface_ppt_for_ice", "surface_ppt_for_snow",
                                         "surface_ppt_for_graupel", "total_surface_ppt"]
    assert all(e["form"] == "scalar" and e["dim"] == "time" and e["units"] == "kg/kg m" for e in tail)
    np.testing.assert_allclose([e["v"] for e in tail], [rppt[0], rppt[3], rppt[1], rppt[2], rppt.sum()], rtol=1e-8, atol=1e-300)
    assert tail[0]["v"] > 0


def test_rate_diagnostics_mixed_nx3_names_order_values(tmp_path, oracle_mixed):
    """nx > 1, mixed phase: per column, per level, all 36 rates in the order of M:3044-3119 through
    save_dg(k, ii, value, ...); then W:248-303: domain means (1-D form, /nx) followed by the per-column values."""
    nx = 3
    log, inp = _run_dump(tmp_path, nx, 4, "mixed", 3)
    rates = [e for e in log if e["form"] == "ki"]
    assert len(rates) == 36 * 120 * nx and log[: len(rates)] == rates
    for n, e in enumerate(rates):
        col, rem = divmod(n, 36 * 120)
        assert e["i"] == col + 1 and e["k"] == rem // 36 + 1 and e["name"] == RATE_NAMES[rem % 36], (n, e)
        assert e["units"] == "/kg/s" and e["dim"] == "z"
    ppts = []
    for c in range(nx):
        rppt, rref, no_micro = _oracle_rates_for(oracle_mixed, inp[c], False)
        assert not no_micro
        ppts.append(rppt)
        got = np.array([e["v"] for e in rates[c * 4320:(c + 1) * 4320]]).reshape(120, 36).T
        assert np.count_nonzero(got[:30]) > 300                                   # frozen-species processes are active
        _check_rate_values(got, rref, "mixed column %d" % c)
    ppts = np.array(ppts)                                                          # [nx, 4] rain, snow, graupel, ice
    tail = log[len(rates):]
    names = ["surface_ppt_for_rain", "surface_ppt_for_ice", "surface_ppt_for_snow", "surface_ppt_for_graupel", "total_surface_ppt"]
    cols = [ppts[:, 0], ppts[:, 3], ppts[:, 1], ppts[:, 2], ppts.sum(axis=1)]
    assert len(tail) == 2 * 5 * nx and all(e["form"] == "array" and e["dim"] == "time" for e in tail)
    for half, div in ((0, nx), (1, 1)):                                            # means first (W:255-275), then columns (W:283-303)
        for m in range(5):
            blk = tail[(half * 5 + m) * nx:(half * 5 + m + 1) * nx]
            assert [e["name"] for e in blk] == [names[m]] * nx and [e["i"] for e in blk] == [1, 2, 3]
            np.testing.assert_allclose([e["v"] for e in blk], cols[m] / div, rtol=1e-8, atol=1e-300)


def test_no_micro_column_emits_no_rate_diagnostics(tmp_path):
    """A column that leaves through the no_micro return (M:1540) never reaches the save_dg block of M:2962: only the
    five precipitation calls of the adapter remain."""
    log, _ = _run_dump(tmp_path, 1, 2, "dry", 2)
    assert [e["form"] for e in log] == ["scalar"] * 5 and all(e["v"] == 0.0 for e in log)
    log, _ = _run_dump(tmp_path, 1, 2, "warm", 2)
    assert len(log) == 6 * 120 + 5


def test_native_real4_kid_build_in_the_reference_native_arithmetic():
    """KiD's default build (4-byte REAL) with kidmp_arith = 'p32n': the REAL arrays go to the GPU as they are and the
    kernel keeps the reference's own REAL / DOUBLE PRECISION split.  The reference AS SHIPPED ends this case at
    sum(qc) 2.218541e-2, sum(qr) 2.693803e-3, sum(nr) 1.060634e6 (native); its P64 build elsewhere."""
    exe32 = os.path.join(ROOT, "kid_amd", "fortran", "build32", "kid_mini_driver")
    if not os.path.exists(exe32):
        pytest.skip("build32 not built (make -C kid_amd/fortran FFLAGS=-O2 B=build32)")
    res = {}
    for arith in ("p32n", "f32"):
        out = subprocess.run([exe32, "1", "360", "warm", "0", arith], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        for line in out.stdout.splitlines():
            p = line.split()
            if p and p[0] == "KATB":
                res[arith] = np.array([float(x) for x in p[1:5]])
    native = np.array([1.530434, 2.218541e-2, 2.693803e-3, 1.060634e6])
    p64 = np.array([1.530434, 2.218719e-2, 2.694135e-3, 1.060568e6])
    print("KAT-B through the Fortran drop-in:", res)
    assert np.all(np.abs(res["p32n"] / native - 1) < 4e-5), res["p32n"]
    assert np.all(np.abs(res["p32n"][1:] - native[1:]) < 0.5 * np.abs(p64[1:] - native[1:])), res["p32n"]
    assert np.all(np.abs(res["f32"] / native - 1) < 3e-4), res["f32"]
    # the 8-byte build refuses the binary32 arithmetic instead of converting silently
    out = subprocess.run([EXE, "1", "2", "warm", "0", "p32n"], capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "4-byte default REAL" in out.stdout + out.stderr
