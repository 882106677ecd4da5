"""Parity of the HIP path (through the C ABI) against the CPU oracle.  -m gpu."""
import numpy as np
import pytest

import cases
import kat_cases as kc
from parity import OUT, TOL, assert_parity, max_rel

pytestmark = pytest.mark.gpu


def _copy(st):
    return {k: np.ascontiguousarray(v.copy()) for k, v in st.items()}


def _oracle_batch(o, st, dt):
    ref = _copy(st)
    ppt = o.batch_step(ref, dt)
    return ref, ppt


def _gpu_batch(m, st, dt, rates=False):
    got = _copy(st)
    ppt, r = m.batch_step_host(got, dt, want_rates=rates)
    return got, ppt, r


def _check(got, gppt, ref, rppt, tol=TOL, mask=None, max_excluded=0):
    mx, per = max_rel(got, ref, OUT, mask)
    pmx = float(np.max(np.abs(gppt - rppt) / np.maximum(np.abs(rppt), 1e-12)))
    assert mx < tol and pmx < tol, (per, pmx)
    if mask is not None:
        assert int((~mask).sum()) <= max_excluded, int((~mask).sum())
    return mx


TABLES = ["tcg_racg", "tmr_racg", "tcr_gacr", "tmg_gacr", "tnr_racg", "tnr_gacr",
          "tcs_racs1", "tmr_racs1", "tcs_racs2", "tmr_racs2", "tcr_sacr1", "tms_sacr1", "tcr_sacr2", "tms_sacr2",
          "tnr_racs1", "tnr_racs2", "tnr_sacr1", "tnr_sacr2", "tpi_qcfz", "tni_qcfz", "tpi_qrfz", "tpg_qrfz",
          "tni_qrfz", "tnr_qrfz", "tps_iaus", "tni_iaus", "tpi_ide", "t_Efrw", "t_Efsw", "tnc_wev"]


@pytest.mark.parametrize("name", TABLES)
def test_table_matches_oracle(gpu_mixed, oracle_mixed, name):
    ref = oracle_mixed.table(name)
    got = gpu_mixed.table(name, ref.shape)
    scale = np.max(np.abs(ref))
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30 + 1e-13 * scale)
    assert np.max(err) < 1e-11, (name, float(np.max(err)))


def test_constants_match_oracle(gpu_mixed, oracle_mixed):
    for n in ("cre", "crg", "cse", "csg", "cge", "cgg", "cie", "cig", "ccg1", "ccg2", "ccg3", "ocg1", "ocg2",
              "Dr", "dtr", "Ds", "dts", "Dg", "dtg", "Di", "dti", "Dc", "t_Nc", "t1_qr_qc", "t2_qr_qi", "t1_qg_qc",
              "t2_qr_ev", "t2_qs_sd", "t1_qs_me", "t2_qs_me", "t2_qg_sd", "t1_qg_me", "t2_qg_me", "Sc3", "D0i",
              "xm0s", "xm0g", "ocms", "ocmg"):
        np.testing.assert_allclose(gpu_mixed.const(n), oracle_mixed.const(n), rtol=1e-15, atol=0, err_msg=n)


def test_interleaved_records_equal_planar_tables(gpu_mixed):
    racs = gpu_mixed.table("racs_rec").reshape(-1, 10)
    for i, n in enumerate(["tmr_racs1", "tcr_sacr1", "tmr_racs2", "tcr_sacr2", "tcs_racs1", "tms_sacr1",
                           "tnr_racs1", "tnr_racs2", "tnr_sacr1", "tnr_sacr2"]):
        assert np.array_equal(racs[:, i], gpu_mixed.table(n)), n
    racg = gpu_mixed.table("racg_rec").reshape(-1, 5)
    for i, n in enumerate(["tmr_racg", "tcr_gacr", "tnr_racg", "tnr_gacr", "tcg_racg"]):
        assert np.array_equal(racg[:, i], gpu_mixed.table(n)), n
    q = gpu_mixed.table("qrfz_rec").reshape(-1, 4)
    for i, n in enumerate(["tpg_qrfz", "tpi_qrfz", "tni_qrfz", "tnr_qrfz"]):
        assert np.array_equal(q[:, i], gpu_mixed.table(n)), n


def test_kat_a_warm_one_step(gpu_warm, oracle_warm):
    st = cases.replicate(kc.kat_a(False), 3)
    ref, rppt = _oracle_batch(oracle_warm, st, 10.0)
    got, gppt, _ = _gpu_batch(gpu_warm, st, 10.0)
    _check(got, gppt, ref, rppt)


def test_kat_a_mixed_one_step(gpu_mixed, oracle_mixed):
    st = cases.replicate(kc.kat_a(True), 2)
    ref, rppt = _oracle_batch(oracle_mixed, st, 10.0)
    got, gppt, _ = _gpu_batch(gpu_mixed, st, 10.0)
    _check(got, gppt, ref, rppt)


def test_edge_cases_one_step(gpu_mixed, oracle_mixed):
    st = cases.edge_cases()
    ref, rppt = _oracle_batch(oracle_mixed, st, 10.0)
    got, gppt, _ = _gpu_batch(gpu_mixed, st, 10.0)
    # levels on the M:3587/M:3596 residue tests must equal one of their two outcomes (see parity.py)
    v = assert_parity(oracle_mixed, st, 10.0, got, gppt)
    assert v["n_branch_levels"] <= 40, v
    assert got["qc"][1, 5] == 0.0           # no_micro early exit still zeroes sub-R1 species (M:1412)


def test_rates_match_oracle(gpu_mixed, oracle_mixed):
    st = cases.edge_cases()
    got, gppt, rates = _gpu_batch(gpu_mixed, st, 10.0, rates=True)
    for c in range(st["qv"].shape[0]):
        col = {k: st[k][c].copy() for k in st}
        _, rref, _, _ = oracle_mixed.column_step(col, 10.0, want_rates=True)
        scale = np.maximum(np.max(np.abs(rref), axis=1, keepdims=True), 1e-300)
        err = np.abs(rates[c] - rref) / np.maximum(np.abs(rref), 1e-9 * scale)
        assert np.max(err) < 1e-10, (c, float(np.max(err)), int(np.argmax(np.max(err, axis=1))))


def test_rates_match_oracle_warm(gpu_warm, oracle_warm):
    """The warm-rain instantiation with the rate buffer: the 6 warm rates of M:3104-3119, zeros for the other 30."""
    st = cases.replicate(kc.kat_a(False), 3)
    st["qr"][1] *= 3.0
    st["qc"][2] *= 0.3
    got, gppt, rates = _gpu_batch(gpu_warm, st, 10.0, rates=True)
    for c in range(3):
        col = {k: st[k][c].copy() for k in st}
        _, rref, _, _ = oracle_warm.column_step(col, 10.0, want_rates=True)
        scale = np.maximum(np.max(np.abs(rref), axis=1, keepdims=True), 1e-300)
        err = np.abs(rates[c] - rref) / np.maximum(np.abs(rref), 1e-9 * scale)
        assert np.max(err) < 1e-10, (c, float(np.max(err)), int(np.argmax(np.max(err, axis=1))))
        assert np.count_nonzero(rates[c][:30]) == 0 and np.count_nonzero(rates[c][30:]) > 0


def test_kat_c_sedimentation_substeps(gpu_mixed, oracle_mixed):
    import torch
    st = cases.replicate(kc.kat_c(), 2)
    ref = _copy(st)
    dev = {k: torch.from_numpy(st[k]).cuda() for k in st}
    ppt = torch.zeros(2, 4, dtype=torch.float64, device="cuda")
    nstep = torch.zeros(2, 4, dtype=torch.int32, device="cuda")
    for n in range(3):
        ppt.zero_()
        gpu_mixed.batch_step(dev, 10.0, ppt, nstep=nstep)
        rppt = oracle_mixed.batch_step(ref, 10.0)
        torch.cuda.synchronize()
        got = {k: dev[k].cpu().numpy() for k in OUT}
        _check(got, ppt.cpu().numpy(), ref, rppt, tol=1e-9)
        assert nstep.cpu().numpy()[0].tolist() == [25 - n, 1, 1, 25 - n]     # KAT-C of SURVEY 9h


def test_config3_sample_one_step(gpu_mixed, oracle_mixed):
    st = cases.config3(ncol=512)
    ref, rppt = _oracle_batch(oracle_mixed, st, 10.0)
    got, gppt, _ = _gpu_batch(gpu_mixed, st, 10.0)
    # this profile keeps liquid cloud up to 228 K in subsaturated air, so ~4 % of its levels sit on the
    # M:3596 residue branch (chaotic in the reference itself, see parity.py)
    assert_parity(oracle_mixed, st, 10.0, got, gppt, max_branch_frac=6e-2)


def test_config5_sample_one_step(gpu_mixed, oracle_mixed):
    st = cases.config5(ncol=256)
    ref, rppt = _oracle_batch(oracle_mixed, st, 10.0)
    got, gppt, _ = _gpu_batch(gpu_mixed, st, 10.0)
    assert_parity(oracle_mixed, st, 10.0, got, gppt, max_branch_frac=2e-3)


def test_config2_warm_replicated(gpu_warm, oracle_warm):
    st = cases.config2(ncol=64)
    ref, rppt = _oracle_batch(oracle_warm, st, 10.0)
    got, gppt, _ = _gpu_batch(gpu_warm, st, 10.0)
    _check(got, gppt, ref, rppt)
    assert np.array_equal(got["qr"][0], got["qr"][-1])      # replicas are bit-identical


def test_multi_step_drift_mixed(gpu_mixed, oracle_mixed):
    """200 coupled steps (KAT-A mixed): device-resident state vs the oracle chain."""
    import torch
    st = cases.replicate(kc.kat_a(True), 1)
    ref = _copy(st)
    dev = {k: torch.from_numpy(st[k]).cuda() for k in st}
    ppt = torch.zeros(1, 4, dtype=torch.float64, device="cuda")
    for n in range(200):
        ppt.zero_()
        gpu_mixed.batch_step(dev, 10.0, ppt)
        rppt = oracle_mixed.batch_step(ref, 10.0)
    torch.cuda.synchronize()
    got = {k: dev[k].cpu().numpy() for k in OUT}
    mx, per = max_rel(got, ref, OUT)
    assert mx < 1e-8, per                                   # drift bound; 1-step bound is TOL
    np.testing.assert_allclose(ppt.cpu().numpy(), rppt, rtol=1e-9, atol=1e-15)
    # and the survey's known answer for the reference itself (SURVEY 9h KAT-A mixed, 6 digits)
    assert abs(got["qs"].sum() / 4.26247e-2 - 1) < 2e-6
    assert abs(float(ppt[0, 0]) / 1.708898e-2 - 1) < 2e-6


def test_table_cache_roundtrip_through_reference_format(gpu_mixed, tmp_path):
    """kidmp_save_table_cache -> run_data-style text files -> kidmp_load_table_cache into a second context."""
    from kid_amd import ThompsonMP, cache_read_file
    d = str(tmp_path)
    gpu_mixed.save_table_cache(d)
    racg = cache_read_file(d + "/racg_thompson09.data", 6, 28 * 28 * 37 * 37)
    assert np.array_equal(racg[1], gpu_mixed.table("tmr_racg"))          # file order of M:3823-3828
    other = ThompsonMP(iiwarm=False)
    try:
        other.load_table_cache(d)
        for n in ("tcg_racg", "tnr_gacr", "tcs_racs1", "tms_sacr2", "racs_rec", "racg_rec"):
            assert np.array_equal(other.table(n), gpu_mixed.table(n)), n
        st = cases.config3(ncol=64)
        a, pa, _ = _gpu_batch(gpu_mixed, st, 10.0)
        b, pb, _ = _gpu_batch(other, st, 10.0)
        for k in OUT:
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(pa, pb)
    finally:
        other.close()


def test_effective_radii_match_oracle(gpu_mixed, oracle_mixed):
    """calc_effectRad (M:4834-4935) on the device vs the oracle: cloud water, cloud ice, snow; absent species keep the preset."""
    import torch
    st = {k: np.concatenate([cases.config3(48)[k], cases.config5(48)[k], cases.edge_cases()[k]]) for k in cases.KEYS}
    ref = oracle_mixed.calc_effectRad(st)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
    got = gpu_mixed.effective_radii(dev)
    for g, r, name, preset in zip(got, ref, ("re_qc", "re_qi", "re_qs"), (2.49e-6, 4.99e-6, 9.99e-6)):
        g = g.cpu().numpy()
        assert np.max(np.abs(g - r) / r) < 1e-12, name
        assert (g == preset).any() and (g != preset).any(), name


# ---- aerosol-aware branch (is_aerosol_aware = .true.): GPU vs the oracle's restatement of M:1410,
#      M:2043-2111, M:2397-2408, M:2602, M:2796-2852, M:2867.  The survey recorded no reference output with the switch
#      on (KiD never sets it), so this branch is "parity unpinned" against the reference; what is held here is that the
#      two restatements agree to the same bound as the rest of the scheme.
def _aero_batch(name, ncol):
    st = getattr(cases, name)(ncol) if name != "edge" else cases.edge_cases()
    rng = np.random.default_rng(5)
    n = st["qv"].shape[0]
    st["w"] = np.ascontiguousarray(rng.uniform(-0.5, 8.0, size=(n, 1)) * np.ones_like(st["qv"]))     # updrafts for activ_ncloud
    st["nwfa"] = st["nwfa"] * rng.uniform(0.3, 30.0, size=(n, 1))                                     # 3e6 ... 3e8 /kg CCN
    st["nifa"] = st["nifa"] * rng.uniform(0.5, 200.0, size=(n, 1))
    st["nc"] = st["nc"] * rng.uniform(0.2, 3.0, size=st["nc"].shape)                                  # a prognostic droplet number
    return {k: np.ascontiguousarray(v) for k, v in st.items()}


@pytest.mark.parametrize("name,ncol", [("config3", 256), ("config5", 128), ("edge", 9)])
def test_aerosol_aware_one_step(gpu_mixed_aero, oracle_mixed_aero, name, ncol):
    st = _aero_batch(name, ncol)
    got, gppt, _ = _gpu_batch(gpu_mixed_aero, st, 10.0)
    v = assert_parity(oracle_mixed_aero, st, 10.0, got, gppt, max_branch_frac=0.1)
    print("aerosol-aware", name, v)
    # the switch does something: the same input through the default context gives another droplet / ice number
    ref_off = _copy(st)
    from oracle.oracle import Oracle
    o_off = Oracle(iiwarm=False)
    o_off.batch_step(ref_off, 10.0)
    o_off.close()
    assert np.max(np.abs(got["nc"] - ref_off["nc"])) > 1.0 and not np.array_equal(got["nwfa"], ref_off["nwfa"])


def test_aerosol_aware_twenty_steps(gpu_mixed_aero, oracle_mixed_aero):
    import torch
    st = _aero_batch("config3", 32)
    ref = _copy(st)
    dev = {k: torch.from_numpy(st[k]).cuda() for k in st}
    ppt = torch.zeros(32, 4, dtype=torch.float64, device="cuda")
    for _ in range(20):
        ppt.zero_()
        gpu_mixed_aero.batch_step(dev, 10.0, ppt)
        rppt = oracle_mixed_aero.batch_step(ref, 10.0)
    torch.cuda.synchronize()
    got = {k: dev[k].cpu().numpy() for k in OUT}
    e = np.concatenate([(np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), 1e4 * {"t": 1e-4}.get(k, 1e-12 if k[0] == "q" else 1e-6))).ravel() for k in OUT])
    assert np.quantile(e, 0.999) < 1e-8, float(np.quantile(e, 0.999))       # coupled drift; chaotic-branch levels may differ
    np.testing.assert_allclose(ppt.cpu().numpy(), rppt, rtol=1e-8, atol=1e-14)


def test_aerosol_aware_context_requires_w(gpu_mixed_aero):
    from kid_amd import KidmpError
    st = _aero_batch("config3", 4)
    st.pop("w")
    import torch
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    with pytest.raises(KidmpError):
        gpu_mixed_aero.batch_step(dev, 10.0, torch.zeros(4, 4, dtype=torch.float64, device="cuda"))
