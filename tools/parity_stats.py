#!/usr/bin/env python
"""Parity statistics of the HIP path vs the CPU oracle on a large sample (SURVEY 9e asks for more than a max):
per variable the max and the 99.9th percentile of |x-ref|/max(|ref|,floor) after ONE step from identical
inputs, the share of levels excluded as chaotic in the reference, and the share of columns whose every
conditioned level is within 1e-10."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from kid_amd import ThompsonMP  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from parity import FLOORS, OUT, TOL, conditioned_mask  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=20000)
    args = ap.parse_args()
    for name, warm in (("config2", True), ("config3", False), ("config5", False)):
        st = getattr(cases, name)(args.ncol if name != "config2" else 2000)
        o, m = Oracle(iiwarm=warm), ThompsonMP(iiwarm=warm)
        ref = {k: v.copy() for k, v in st.items()}
        rppt = o.batch_step(ref, 10.0)
        mask = conditioned_mask(o, st, 10.0, ref)
        got = {k: v.copy() for k, v in st.items()}
        gppt, _ = m.batch_step_host(got, 10.0)
        per, worst_col = {}, np.zeros(st["qv"].shape[0])
        for k in OUT:
            e = np.where(mask, np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), FLOORS[k]), 0.0)
            per[k] = {"max": float(e.max()), "q999": float(np.quantile(e, 0.999))}
            worst_col = np.maximum(worst_col, e.max(axis=1))
        pe = np.abs(gppt - rppt) / np.maximum(np.abs(rppt), 1e-12)
        print(json.dumps({"workload": name, "columns": int(st["qv"].shape[0]), "per_variable": per,
                          "precip_max_rel": float(pe.max()),
                          "levels_excluded_frac": float((~mask).mean()),
                          "columns_within_1e-10_frac": float((worst_col < TOL).mean()),
                          "max_rel_overall": float(worst_col.max())}))
        o.close(); m.close()


if __name__ == "__main__":
    main()
