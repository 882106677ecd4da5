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
    # the full ordered name set of the adapter for nx > 1 (W:248-307), 'total_ppt_level' (dim='z,x', shape (nz, nx)) last
    level = tail[2 * 5 * nx:]
    tail = tail[: 2 * 5 * nx]
    seen = []
    for e in tail + level:
        if not seen or seen[-1] != e["name"]:
            seen.append(e["name"])
    assert seen == names * 2 + ["total_ppt_level"], seen
    assert len(level) == 120 * nx and all(e["form"] == "2d" and e["dim"] == "z,x" and e["units"] == "kg/kg m" for e in level)
    assert [(e["i"], e["k"]) for e in level] == [(i + 1, k + 1) for i in range(nx) for k in range(120)]
    assert all(e["v"] == 0.0 for e in level)          # the reference never assigns the array (W:191): defined as zeros here
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
    # (this driver is default REAL like the survey's native probe: it forms the inputs in binary32 itself, and the Python
    #  twin of it, kat_cases.kat_b_native through tests/test_gpu_precision.py, ends on the same seven digits)
    assert np.all(np.abs(res["p32n"] / native - 1) < 2.5e-5), res["p32n"]      # measured 4e-7, 4e-6, 1.5e-5, 1.1e-5
    assert np.all(np.abs(res["p32n"][1:] - native[1:]) < 0.5 * np.abs(p64[1:] - native[1:])), res["p32n"]
    assert np.all(np.abs(res["f32"] / native - 1) < 3e-4), res["f32"]
    # the 8-byte build refuses the binary32 arithmetic instead of converting silently
    out = subprocess.run([EXE, "1", "2", "warm", "0", "p32n"], capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "4-byte default REAL" in out.stdout + out.stderr


# ---------------------------------------------------------------------------------------------------------------
# The adapter's forcing path (W:59-97 forms state + (adv + div)*dt, W:198-245 subtracts adv + div again), the warm
# aerosol-aware call, the reference's table-cache switch and the multi-device entry, all through the Fortran drop-in.
_ENV = dict(os.environ, OMP_NUM_THREADS="8")
_HYD = ((0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (4, 0))     # (KiD species - 1, moment - 1) in dump order


def _run_opts(tmp_path, nx, nsteps, case, *opts, dump_step=0):
    out = subprocess.run([EXE, str(nx), str(nsteps), case, str(dump_step), "p64", "-"] + list(opts), capture_output=True,
                         text=True, timeout=900, cwd=str(tmp_path), env=_ENV)
    assert out.returncode == 0, out.stdout + out.stderr
    vals = {}
    for line in out.stdout.splitlines():
        p = line.split()
        if p and p[0] in ("KATB", "KATBN", "PSUM", "TIME"):
            vals[p[0]] = np.array([float(x) for x in p[1:]])
    return vals, out.stdout


def _oracle_adapter(oracle, d, nx, nz=120, dt=10.0):
    """oracle.kid_interface on the dumped inputs; returns (dtheta, dqv, dhyd[7]) as [nx, nz] arrays."""
    def f(a):                                                         # [nx, nz] -> Fortran (nz, nx) flattened
        return np.ascontiguousarray(a).ravel()
    hy, ha, hd = (np.zeros((2, 5, nx, nz)) for _ in range(3))         # hydro[k + nz*(i + nx*(ih + 5*imom))]
    for n, (ih, im) in enumerate(_HYD):
        hy[im, ih], ha[im, ih], hd[im, ih] = d[:, :, 4 + n], d[:, :, 15 + n], d[:, :, 22 + n]
    dth, dqv, dhy, _ = oracle.kid_interface(nz, nx, dt, 1.0e5, 287.058 / 1005.0, f(d[:, :, 0]), f(d[:, :, 11]), f(d[:, :, 12]),
                                            f(d[:, :, 1]), d[0, :, 3].copy(), f(d[:, :, 2]), f(d[:, :, 13]), f(d[:, :, 14]),
                                            hy.ravel(), ha.ravel(), hd.ravel())
    dhy = dhy.reshape(2, 5, nx, nz)
    return dth.reshape(nx, nz), dqv.reshape(nx, nz), [dhy[im, ih] for ih, im in _HYD]


def _check_mphys(d, ref, nx, what):
    dth, dqv, dhy = ref
    got = [d[:, :, 29], d[:, :, 30]] + [d[:, :, 31 + n] for n in range(7)]
    want = [dth, dqv] + dhy
    # tendencies are differences of O(1) states divided by dt: measure against the size of the state's own rate
    scale = [np.abs(d[:, :, 0]).max() / 10.0, np.abs(d[:, :, 2]).max() / 10.0] + [max(np.abs(d[:, :, 4 + n]).max() / 10.0, 1e-300) for n in range(7)]
    for n, (g, w, sc) in enumerate(zip(got, want, scale)):
        err = np.abs(g - w) / np.maximum(np.abs(w), 1e-5 * sc)      # floor: a few ulp of state / dt
        assert err.max() < 1e-9, (what, n, float(err.max()))
    return got, want


@pytest.mark.parametrize("case", ["warm", "mixed"])
def test_adapter_forcing_terms_match_oracle(tmp_path, oracle_warm, oracle_mixed, case):
    """A prescribed updraft + divergence forcing (non-zero dtheta_adv/div, dqv_adv/div, dhydrometeors_adv/div): the
    drop-in's d*_mphys equal the oracle adapter's (W:59-97, W:198-245) at 1e-9, and the forcing really acts."""
    nx = 3
    _run_opts(tmp_path, nx, 6, case, "forcing=1", "mphys=5")
    d = np.loadtxt(os.path.join(str(tmp_path), "mphys_dump.txt")).reshape(nx, 120, 38)
    assert np.abs(d[:, :, 11]).max() > 1e-4 and np.abs(d[:, :, 13]).max() > 1e-8 and np.abs(d[:, :, 15:29]).max() > 1e-9
    oracle = oracle_warm if case == "warm" else oracle_mixed
    got, want = _check_mphys(d, _oracle_adapter(oracle, d, nx), nx, case)
    # with the forcing terms dropped from the oracle's input the tendencies differ visibly: the terms are not inert
    d0 = d.copy()
    d0[:, :, 11:29] = 0.0
    dth0, _, _ = _oracle_adapter(oracle, d0, nx)
    assert np.abs(dth0 - want[0]).max() > 1e-6


def test_warm_aerosol_aware_run_through_the_drop_in(tmp_path):
    """iiwarm with is_aerosol_aware: the adapter passes all twelve state slots, the four frozen ones zeroed (W:46-52);
    d*_mphys equal the oracle adapter's with the same switches."""
    from oracle.oracle import Oracle
    nx = 2
    _run_opts(tmp_path, nx, 5, "warm", "aero=1", "mphys=5")
    d = np.loadtxt(os.path.join(str(tmp_path), "mphys_dump.txt")).reshape(nx, 120, 38)
    o = Oracle(iiwarm=True, aerosol_aware=True)
    try:
        _check_mphys(d, _oracle_adapter(o, d, nx), nx, "warm aerosol-aware")
    finally:
        o.close()
    assert np.all(d[:, :, 34:38] == 0.0)                               # no frozen-species tendency out of a warm run


def test_l_reuse_thompson_lookup_semantics(tmp_path, gpu_mixed):
    """The reference's cache switch (M:3717-3729, M:3864-3895) in the drop-in: with the switch set a file that exists is
    READ (here: tables of zeros, so collection of rain by snow / graupel vanishes and the run ends elsewhere); with the
    switch off the same files are ignored and OVERWRITTEN by the freshly built tables; a later run with the switch set
    then reproduces the computed-table run bit for bit (17 significant digits round-trip binary64)."""
    from kid_amd import cache_read_file, cache_write_file
    rd = tmp_path / "run_data"
    rd.mkdir()
    plain = tmp_path / "plain"
    plain.mkdir()
    base, _ = _run_opts(plain, 2, 3, "mixed")                          # no run_data beside this run: nothing read or written
    assert not list(plain.glob("run_data*"))
    n_racg, n_racs = 37 * 37 * 28 * 28, 28 * 9 * 37 * 37
    cache_write_file(str(rd / "racg_thompson09.data"), [np.zeros(n_racg)] * 6)
    cache_write_file(str(rd / "racs_thompson09.data"), [np.zeros(n_racs)] * 12)
    zeroed, out = _run_opts(tmp_path, 2, 3, "mixed", "reuse=1")
    assert "Reading in pre-calculated lookup tables" in out
    assert not np.array_equal(zeroed["KATB"], base["KATB"])           # the perturbed tables were picked up
    ignored, out = _run_opts(tmp_path, 2, 3, "mixed", "reuse=0")
    assert "Reading in pre-calculated" not in out
    assert np.array_equal(ignored["KATB"], base["KATB"]) and np.array_equal(ignored["KATBN"], base["KATBN"])
    got = cache_read_file(str(rd / "racg_thompson09.data"), 6, n_racg)   # ... and replaced by the built tables
    for name, g in zip(("tcg_racg", "tmr_racg", "tcr_gacr", "tmg_gacr", "tnr_racg", "tnr_gacr"), got):
        assert np.array_equal(g, gpu_mixed.table(name)), name
    again, out = _run_opts(tmp_path, 2, 3, "mixed", "reuse=1")
    assert "Reading in pre-calculated lookup tables" in out
    assert np.array_equal(again["KATB"], base["KATB"]) and np.array_equal(again["KATBN"], base["KATBN"])


def test_two_devices_through_the_drop_in(tmp_path):
    """kidmp_devices = (0, 0): the columns are spread over two contexts by kidmp_batch_step_host_multi; end state
    bitwise as on one device, and the RCCL-reduced domain sums equal the sum of the per-column precipitation the adapter
    hands to save_dg (W:283-303)."""
    nx = 7
    one, _ = _run_opts(tmp_path, nx, 60, "warm")
    two, _ = _run_opts(tmp_path, nx, 60, "warm", "devices=0,0", dump_step=60)
    assert np.array_equal(one["KATB"], two["KATB"]) and np.array_equal(one["KATBN"], two["KATBN"])
    assert "PSUM" in two and "PSUM" not in one
    per_col = {}
    for line in open(os.path.join(str(tmp_path), "dg_dump.txt")):
        p = line.split()
        if p[0] == "array":
            per_col.setdefault(p[1], []).append(float(p[4]))
    rain = np.array(per_col["surface_ppt_for_rain"][nx:])             # second block: the per-column values
    assert rain.sum() > 0
    np.testing.assert_allclose(two["PSUM"][0], rain.sum(), rtol=1e-13)
    assert np.all(two["PSUM"][1:] == 0.0)
