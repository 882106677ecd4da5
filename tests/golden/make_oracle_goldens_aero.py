"""Generates tests/golden/oracle_goldens_aero.npz: 17-digit input/output vectors of the CPU oracle with
is_aerosol_aware = .true. (M:28) -- 9 edge-case columns + 7 config-3 columns with random updrafts and aerosol loads, one
mp_thompson call (dt = 10 s), plus calc_effectRad on the inputs.  Freezes the oracle's aerosol-aware branch (which has
no reference output to be pinned on) against silent drift, and gives the GPU tests a committed target.
    python tests/golden/make_oracle_goldens_aero.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def inputs():
    ec, c3 = cases.edge_cases(), cases.config3(7)
    st = {k: np.concatenate([ec[k], c3[k]]) for k in cases.KEYS}
    rng = np.random.default_rng(2024)
    n = st["qv"].shape[0]
    st["w"] = rng.uniform(-0.5, 8.0, size=(n, 1)) * np.ones_like(st["qv"])
    st["nwfa"] = st["nwfa"] * rng.uniform(0.3, 30.0, size=(n, 1))
    st["nifa"] = st["nifa"] * rng.uniform(0.5, 200.0, size=(n, 1))
    st["nc"] = st["nc"] * rng.uniform(0.2, 3.0, size=st["nc"].shape)
    return {k: np.ascontiguousarray(v) for k, v in st.items()}


if __name__ == "__main__":
    o = Oracle(iiwarm=False, aerosol_aware=True)
    st = inputs()
    out = {"in_" + k: v.copy() for k, v in st.items()}
    re = o.calc_effectRad(st)
    work = {k: v.copy() for k, v in st.items()}
    ppt, flags = o.batch_step(work, 10.0, want_illcond=True)
    out.update({"out_" + k: work[k] for k in cases.KEYS})
    out.update(ppt=ppt, flags=flags, re_qc=re[0], re_qi=re[1], re_qs=re[2])
    path = os.path.join(ROOT, "tests", "golden", "oracle_goldens_aero.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.0f KB" % (os.path.getsize(path) / 1024))
