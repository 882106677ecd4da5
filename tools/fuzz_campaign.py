#!/usr/bin/env python3
"""Differential fuzz campaign (run on the GPU box): tests/test_gpu_fuzz.py's generator on many seeds, level counts and time steps,
HIP path against the CPU oracle with the branch-aware comparison of tests/parity.py.  One JSON line per seed
(profiles/rNN_fuzz_campaign.jsonl) and a last line with the totals.

    python tools/fuzz_campaign.py [--seeds 40] [--ncol 1000] > profiles/rNN_fuzz_campaign.jsonl

Per seed: worst = the largest level error of the batch (levels on the reference's residue-decided tests against the better of
their two outcomes), cols_within_tol = columns whose every level is within max(1e-10, 10 x the oracle's own ulp-sensitivity
there), levels_gt_1e7 / levels_gt_1e5 = levels further off than 1e-7 / 1e-5, levels_beyond = levels beyond
max(1e-10, 10 x sensitivity), q999 = the 99.9 % quantile of the level errors (no allowance of any kind)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from kid_amd import ThompsonMP
from oracle.oracle import Oracle
from parity import branch_aware_compare
from test_gpu_fuzz import fuzz_columns

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=40)
ap.add_argument("--ncol", type=int, default=1000)
ap.add_argument("--first-seed", type=int, default=100)
ap.add_argument("--warm-seeds", type=int, default=10, help="additional seeds through the warm-rain kernel (iiwarm contexts): even ones "
                "with the frozen species zeroed (the normal KiD warm case), odd ones with ice, snow and graupel present")
ap.add_argument("--lib", default=None, help="another build of libkidmp.so (A/B of the parity tail between kernels)")
args = ap.parse_args()
if args.lib:
    import kid_amd
    kid_amd.load_library(args.lib)

NZ = (64, 128, 120, 77, 120, 200, 120, 33)
DT = (5.0, 10.0, 10.0, 10.0, 2.0, 10.0, 20.0, 10.0)
m, o = ThompsonMP(iiwarm=False), Oracle(iiwarm=False, nthreads=min(os.cpu_count() or 1, 16))
tot = dict(columns=0, levels=0, columns_with_a_level_beyond=0, levels_beyond=0, levels_gt_1e7=0, levels_gt_1e5=0, worst=0.0)
for i in range(args.seeds):
    seed = args.first_seed + i
    nz, dt = NZ[i % len(NZ)], DT[i % len(DT)]
    st = fuzz_columns(args.ncol, nz, seed)
    got = {k: v.copy() for k, v in st.items()}
    gppt, _ = m.batch_step_host(got, dt)
    cmp = branch_aware_compare(o, st, dt, got, gppt, depletion=1e-5)
    err, sens = cmp["err"], cmp["sens"]
    lim = np.maximum(1e-10, 10.0 * sens)
    beyond = err > lim
    rec = dict(seed=seed, nz=nz, dt=dt, worst=float(err.max()), cols_within_tol=float((~beyond).all(axis=1).mean()),
               levels_beyond=int(beyond.sum()), levels_gt_1e7=int((err > 1e-7).sum()), levels_gt_1e5=int((err > 1e-5).sum()),
               q999=float(np.quantile(err, 0.999)), branch_frac=float((cmp["flags"] != 0).mean()),
               precip_worst=float(cmp["ppt_err"].max()))
    print(json.dumps(rec), flush=True)
    tot["columns"] += args.ncol; tot["levels"] += int(err.size)
    tot["columns_with_a_level_beyond"] += int(beyond.any(axis=1).sum()); tot["levels_beyond"] += rec["levels_beyond"]
    tot["levels_gt_1e7"] += rec["levels_gt_1e7"]; tot["levels_gt_1e5"] += rec["levels_gt_1e5"]; tot["worst"] = max(tot["worst"], rec["worst"])
tot["kernel_fingerprint"] = m.kernel_fingerprint("p64")
print(json.dumps(dict(total=tot)), flush=True)
m.close(); o.close()

if args.warm_seeds:
    m, o = ThompsonMP(iiwarm=True), Oracle(iiwarm=True, nthreads=min(os.cpu_count() or 1, 16))
    wt = dict(columns=0, levels=0, columns_with_a_level_beyond=0, levels_beyond=0, levels_gt_1e7=0, worst=0.0)
    for i in range(args.warm_seeds):
        seed = args.first_seed + 1000 + i
        nz, dt = NZ[i % len(NZ)], DT[i % len(DT)]
        st = fuzz_columns(args.ncol, nz, seed)
        if i % 2 == 0:
            for k in ("qi", "ni", "qs", "qg"):
                st[k][:] = 0.0
        got = {k: v.copy() for k, v in st.items()}
        gppt, _ = m.batch_step_host(got, dt)
        cmp = branch_aware_compare(o, st, dt, got, gppt, depletion=1e-5)
        err, sens = cmp["err"], cmp["sens"]
        beyond = err > np.maximum(1e-10, 10.0 * sens)
        rec = dict(warm_seed=seed, nz=nz, dt=dt, frozen_species_present=bool(i % 2), worst=float(err.max()),
                   cols_within_tol=float((~beyond).all(axis=1).mean()), levels_beyond=int(beyond.sum()),
                   levels_gt_1e7=int((err > 1e-7).sum()), q999=float(np.quantile(err, 0.999)), precip_worst=float(cmp["ppt_err"].max()))
        print(json.dumps(rec), flush=True)
        wt["columns"] += args.ncol; wt["levels"] += int(err.size); wt["columns_with_a_level_beyond"] += int(beyond.any(axis=1).sum())
        wt["levels_beyond"] += rec["levels_beyond"]; wt["levels_gt_1e7"] += rec["levels_gt_1e7"]; wt["worst"] = max(wt["worst"], rec["worst"])
    wt["kernel_fingerprint"] = m.kernel_fingerprint("p64")
    print(json.dumps(dict(total_warm=wt)), flush=True)
    m.close(); o.close()
