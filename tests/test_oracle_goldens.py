"""The oracle against its own committed 17-digit fixtures (tests/golden/oracle_goldens.npz, written by
tests/golden/make_oracle_goldens.py after the oracle passed the survey's known answers): an edit of oracle/*.c
that moves any output of the edge-case columns, the KAT first steps, the 36 rates or the KiD adapter shows here.
Agreement is demanded to 1e-12 rather than bitwise only because glibc picks FMA / non-FMA variants of its
exp/log/pow by CPU model, which may differ in the last bit between hosts."""
import os

import numpy as np
import pytest

import cases
import kat_cases as kc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_goldens.npz"))
KEYS = cases.KEYS
TOL = 1e-12


def _rel(a, b, floor):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def _check(o, tag):
    from parity import FLOORS
    st = {k: G["%s_in_%s" % (tag, k)].copy() for k in KEYS}
    ncol = st["qv"].shape[0]
    for c in range(ncol):
        col = {k: np.ascontiguousarray(st[k][c]) for k in KEYS}
        ppt, rates, ns, _ = o.column_step(col, 10.0, want_rates=True)
        for k in KEYS:
            assert _rel(col[k], G["%s_out_%s" % (tag, k)][c], FLOORS.get(k, 1e-300)) < TOL, (tag, c, k)
        assert _rel(ppt, G[tag + "_ppt"][c], 1e-12) < TOL, (tag, c)
        ref = G[tag + "_rates"][c]
        scale = np.maximum(np.max(np.abs(ref), axis=1, keepdims=True), 1e-300)
        assert float(np.max(np.abs(rates - ref) / np.maximum(np.abs(ref), 1e-9 * scale))) < 1e-10, (tag, c)
        assert list(ns) == G[tag + "_nstep"][c].tolist(), (tag, c)


def test_fixture_inputs_are_the_documented_recipes():
    ec = cases.edge_cases()
    for k in KEYS:
        assert np.array_equal(G["edge_in_" + k], ec[k]), k
        assert np.array_equal(G["kata_mixed_in_" + k][0], kc.kat_a(True)[k]), k
        assert np.array_equal(G["katc_in_" + k][0], kc.kat_c()[k]), k


def test_kat_a_warm_first_step(oracle_warm):
    _check(oracle_warm, "kata_warm")


@pytest.mark.slow
def test_edge_cases_kat_a_mixed_kat_c_first_steps(oracle_mixed):
    for tag in ("edge", "kata_mixed", "katc"):
        _check(oracle_mixed, tag)
    _, flags = oracle_mixed.batch_step({k: G["edge_in_" + k].copy() for k in KEYS}, 10.0, want_illcond=True)
    assert np.array_equal(flags, G["edge_flags"])


def test_kid_adapter_first_call(oracle_warm):
    c = kc.kat_b()
    z0, zh = np.zeros(c["nz"]), np.zeros(c["hydro"].size)
    dth, dqv, dhy, ppt = oracle_warm.kid_interface(c["nz"], 1, c["dt"], c["p0"], c["r_on_cp"], c["theta"], z0, z0,
                                                   c["exner"], c["dz"], c["qv"], z0, z0, c["hydro"], zh, zh)
    for got, name, floor in ((dth, "katb_dtheta", 1e-6), (dqv, "katb_dqv", 1e-13), (dhy, "katb_dhydro", 1e-13), (ppt, "katb_ppt", 1e-12)):
        assert _rel(np.asarray(got), G[name], floor) < 1e-10, name


@pytest.mark.slow
def test_aerosol_aware_branch_against_its_fixture():
    """The oracle's aerosol-aware branch (no reference output exists for it) vs its committed vectors."""
    from oracle.oracle import Oracle
    from parity import FLOORS
    A = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_goldens_aero.npz"))
    o = Oracle(iiwarm=False, aerosol_aware=True)
    st = {k: A["in_" + k].copy() for k in KEYS}
    re = o.calc_effectRad(st)
    for got, name in zip(re, ("re_qc", "re_qi", "re_qs")):
        assert _rel(got, A[name], 1e-30) < TOL, name
    ppt, flags = o.batch_step(st, 10.0, want_illcond=True)
    for k in KEYS:
        assert _rel(st[k], A["out_" + k], FLOORS.get(k, 1e-300)) < TOL, k
    assert _rel(ppt, A["ppt"], 1e-12) < TOL and np.array_equal(flags, A["flags"])
    # and the switch matters: the default context gives other droplet numbers from the same input
    o2 = Oracle(iiwarm=False)
    s2 = {k: A["in_" + k].copy() for k in KEYS}
    o2.batch_step(s2, 10.0)
    assert not np.array_equal(s2["nc"], A["out_nc"])
    o.close(); o2.close()
