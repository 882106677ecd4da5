"""Generates tests/golden/config2_column_t900.npz: the BASELINE config-2 base column.

Recipe: the config-1 warm column (KAT-B sounding: dz=25 m, static cloud/rain
layer, zero forcing, iiwarm=T, set_Nc=100) advanced 90 steps of dt=10 s (t=900 s) by the CPU
oracle, with the non-aerosol defaults for nc/nwfa/nifa.  Run from the repo root:
    python tests/golden/make_config2_column.py
The file holds INPUT data for config 2 (15 profiles of 120 float64), nothing else.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

if __name__ == "__main__":
    o = Oracle(iiwarm=True)
    st = cases.warm_column_t0()
    for _ in range(90):
        o.column_step(st, 10.0)
    out = os.path.join(ROOT, "tests", "golden", "config2_column_t900.npz")
    np.savez(out, **st)
    print("wrote", out, {k: float(v.sum()) for k, v in st.items() if k in ("qc", "qr", "nr")})
