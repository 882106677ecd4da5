"""Generates tests/golden/oracle_goldens.npz: 17-digit (binary64) input/output vectors of the CPU oracle.

Why: the oracle is pinned on the reference only through the survey's 6-7 digit known answers
(tests/test_oracle_kat.py).  These fixtures freeze the oracle's full-precision behaviour AFTER it
passes those KATs, so that a later edit of oracle/*.c cannot drift silently (tests/test_oracle_goldens.py),
and give the GPU tests a committed target that does not depend on the oracle being rebuilt
(tests/test_gpu_goldens.py).  They are data: inputs and outputs only.

Contents (all float64; one mp_thompson call, dt = 10 s, unless noted):
  edge_in_<v>, edge_out_<v>      the 9 edge-case columns of tests/cases.py (mixed-phase context)
  edge_ppt, edge_rates, edge_nstep, edge_flags
  kata_warm_* / kata_mixed_* / katc_*   first step of KAT-A warm, KAT-A mixed, KAT-C: in, out, ppt, rates
  katb_*                         first call of the KiD adapter on KAT-B: dtheta, dqv, dhydro, ppt

Run from the repo root (refuses to write unless the KATs pass):
    python tests/golden/make_oracle_goldens.py
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import kat_cases as kc  # noqa: E402
from oracle.oracle import NRATES, Oracle  # noqa: E402

KEYS = cases.KEYS


def one_step(o, cols, tag, out):
    st = {k: np.ascontiguousarray(np.stack([c[k] for c in cols])) for k in KEYS}
    ncol, nz = st["qv"].shape
    for k in KEYS:
        out["%s_in_%s" % (tag, k)] = st[k].copy()
    rates = np.zeros((ncol, NRATES, nz))
    nstep = np.zeros((ncol, 4), dtype=np.int32)
    ppt = np.zeros((ncol, 4))
    work = {k: v.copy() for k, v in st.items()}
    for c in range(ncol):
        col = {k: work[k][c] for k in KEYS}
        p, r, ns, _ = o.column_step(col, 10.0, want_rates=True)
        ppt[c], rates[c], nstep[c] = p, r, ns
    _, flags = o.batch_step({k: v.copy() for k, v in st.items()}, 10.0, want_illcond=True)
    for k in KEYS:
        out["%s_out_%s" % (tag, k)] = work[k]
    out[tag + "_ppt"], out[tag + "_rates"], out[tag + "_nstep"], out[tag + "_flags"] = ppt, rates, nstep, flags


def main():
    rc = subprocess.call([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(ROOT, "tests", "test_oracle_kat.py")])
    if rc != 0:
        raise SystemExit("the oracle does not reproduce the survey's known answers: not writing fixtures")
    out = {}
    om, ow = Oracle(iiwarm=False), Oracle(iiwarm=True)
    ec = cases.edge_cases()
    one_step(om, [{k: ec[k][c] for k in KEYS} for c in range(ec["qv"].shape[0])], "edge", out)
    one_step(ow, [kc.kat_a(False)], "kata_warm", out)
    one_step(om, [kc.kat_a(True)], "kata_mixed", out)
    one_step(om, [kc.kat_c()], "katc", out)
    c = kc.kat_b()
    z0, zh = np.zeros(c["nz"]), np.zeros(c["hydro"].size)
    dth, dqv, dhy, ppt = ow.kid_interface(c["nz"], 1, c["dt"], c["p0"], c["r_on_cp"], c["theta"], z0, z0, c["exner"],
                                          c["dz"], c["qv"], z0, z0, c["hydro"], zh, zh)
    out.update(katb_dtheta=dth, katb_dqv=dqv, katb_dhydro=dhy, katb_ppt=ppt)
    path = os.path.join(ROOT, "tests", "golden", "oracle_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.0f KB" % (len(out), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
