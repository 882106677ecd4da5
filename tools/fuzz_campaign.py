#!/usr/bin/env python
"""Large differential-fuzz campaign (HIP path vs CPU oracle): the generator of tests/test_gpu_fuzz.py over many seeds.
Prints one JSON line per seed and a summary: worst conditioned level, share of columns with every conditioned
level within 1e-10, number of levels beyond 1e-7 / 1e-5 (a branch taken differently would be O(1))."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import fuzz_columns  # noqa: E402
from kid_amd import ThompsonMP  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from parity import FLOORS, OUT, TOL, conditioned_mask  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--ncol", type=int, default=1000)
    args = ap.parse_args()
    o, m = Oracle(iiwarm=False), ThompsonMP(iiwarm=False)
    tot = dict(columns=0, levels=0, excluded=0, cols_bad=0, gt1e7=0, gt1e5=0, worst=0.0, precip_worst=0.0)
    for seed in range(100, 100 + args.seeds):
        nz = (120, 120, 77, 200, 64, 128)[seed % 6]
        dt = (10.0, 10.0, 2.0, 10.0, 5.0, 10.0)[seed % 6]
        st = fuzz_columns(args.ncol, nz, seed)
        ref = {k: v.copy() for k, v in st.items()}
        rppt = o.batch_step(ref, dt)
        mask = conditioned_mask(o, st, dt, ref)
        got = {k: v.copy() for k, v in st.items()}
        gppt, _ = m.batch_step_host(got, dt)
        emax = np.zeros(st["qv"].shape)
        for k in OUT:
            scale = np.maximum(np.maximum(np.abs(ref[k]), FLOORS[k]), 1e-5 * np.abs(st[k]))
            emax = np.maximum(emax, np.where(mask, np.abs(got[k] - ref[k]) / scale, 0.0))
        pe = float(np.max(np.abs(gppt - rppt) / np.maximum(np.abs(rppt), 1e-12)))
        row = dict(seed=seed, nz=nz, dt=dt, worst=float(emax.max()), cols_within_tol=float((emax.max(axis=1) <= TOL).mean()),
                   levels_gt_1e7=int((emax > 1e-7).sum()), levels_gt_1e5=int((emax > 1e-5).sum()),
                   excluded_frac=float((~mask).mean()), precip_worst=pe)
        print(json.dumps(row), flush=True)
        tot["columns"] += args.ncol; tot["levels"] += mask.size; tot["excluded"] += int((~mask).sum())
        tot["cols_bad"] += int((emax.max(axis=1) > TOL).sum()); tot["gt1e7"] += row["levels_gt_1e7"]
        tot["gt1e5"] += row["levels_gt_1e5"]; tot["worst"] = max(tot["worst"], row["worst"])
        tot["precip_worst"] = max(tot["precip_worst"], pe)
    print(json.dumps({"summary": tot}))


if __name__ == "__main__":
    main()
