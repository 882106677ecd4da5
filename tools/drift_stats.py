#!/usr/bin/env python3
"""Drift statistics of BASELINE configs 3 and 5 (SURVEY 8d "drift after the full run"): tests/drift.py on 256-column
samples, 60 and 360 steps; one JSON line per workload and mark.   usage: python tools/drift_stats.py [ncol]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
import drift
from kid_amd import ThompsonMP
from oracle.oracle import Oracle

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m, o = ThompsonMP(iiwarm=False), Oracle(iiwarm=False, nthreads=min(os.cpu_count() or 1, 16))
for name in ("config3", "config5"):
    st = getattr(cases, name)(ncol, seed=cases.SEED + 11)
    for rec in drift.run_chains(m, o, st, 10.0, (60, 360)):
        print(json.dumps(dict(workload=name, **rec)), flush=True)
m.close(); o.close()
