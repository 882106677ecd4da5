#!/usr/bin/env python3
"""Parity statistics of the BASELINE workloads on large samples (run on the GPU box): one step of the HIP path against the CPU
oracle from identical inputs, every level compared (tests/parity.py: levels on the reference's residue-decided tests against the
better of their two outcomes), plus one record per level beyond 1e-10 with the oracle's own response to a 2-4 ulp input change
there.  profiles/rNN_parity_stats.jsonl.

    python tools/parity_stats.py [--ncol 20000] [--seed-offset 7] > profiles/rNN_parity_stats.jsonl"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import cases
from kid_amd import ThompsonMP
from oracle.oracle import Oracle
from parity import FLOORS, OUT, branch_aware_compare, rel_err

ap = argparse.ArgumentParser()
ap.add_argument("--ncol", type=int, default=20000)
ap.add_argument("--seed-offset", type=int, default=7, help="the sample differs from bench.py's accuracy leg (offset 0 is the bench's own)")
args = ap.parse_args()

for name, warm, ncol in (("config2", True, 2000), ("config3", False, args.ncol), ("config5", False, args.ncol)):
    m, o = ThompsonMP(iiwarm=warm), Oracle(iiwarm=warm, nthreads=min(os.cpu_count() or 1, 16))
    st = getattr(cases, name)(ncol) if name == "config2" else getattr(cases, name)(ncol, seed=cases.SEED + args.seed_offset)
    got = {k: v.copy() for k, v in st.items()}
    gppt, _ = m.batch_step_host(got, 10.0)
    cmp = branch_aware_compare(o, st, 10.0, got, gppt)
    err, sens, flags, ref = cmp["err"], cmp["sens"], cmp["flags"], cmp["ref"]
    off = flags == 0
    per = {}
    for v in OUT:
        e = rel_err(got[v], ref[v], FLOORS[v])[off]
        per[v] = dict(max=float(e.max()) if e.size else 0.0, q999=float(np.quantile(e, 0.999)) if e.size else 0.0)
    lim = np.maximum(1e-10, 10.0 * sens)
    print(json.dumps(dict(workload=name, columns=ncol, levels=int(err.size), per_variable_levels_off_the_residue_tests=per,
                          precip_max_rel=float(cmp["ppt_err"].max()), levels_on_residue_tests=int((~off).sum()),
                          levels_matching_neither_outcome_or_beyond_10x_sensitivity=int((err > lim).sum()),
                          max_rel_all_levels_best_outcome=float(err.max()), levels_beyond_1e_10=int((err > 1e-10).sum()),
                          columns_within_1e_10_frac=float((err <= 1e-10).all(axis=1).mean()),
                          columns_within_1e_10_or_10x_sensitivity_frac=float((err <= lim).all(axis=1).mean()),
                          kernel_fingerprint=m.kernel_fingerprint("p64"))), flush=True)
    bad = np.argwhere(err > 1e-10)
    for c, k in bad[np.argsort(-err[err > 1e-10])][:12]:
        pv = {v: float(rel_err(got[v][c, k], ref[v][c, k], FLOORS[v])) for v in OUT}
        w = max(pv, key=pv.get)
        print(json.dumps(dict(tail=name, column=int(c), level=int(k), hip_vs_oracle=float(err[c, k]),
                              oracle_vs_oracle_under_ulp_perturbation=float(sens[c, k]),
                              ratio=float(err[c, k] / max(sens[c, k], 1e-300)), on_residue_test=int(flags[c, k]), worst_variable=w,
                              T=float(st["t"][c, k]), qv=float(st["qv"][c, k]), qc_in=float(st["qc"][c, k]), qc_out=float(ref["qc"][c, k]))),
              flush=True)
    m.close(); o.close()
