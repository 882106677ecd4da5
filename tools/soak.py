#!/usr/bin/env python3
"""Long runs of the column step at BASELINE sizes with nothing but the domain sanity scan watching: every value finite,
no negative mixing ratio or number, maxima within physical bounds, surface precipitation monotone.
usage: python tools/soak.py [steps] [arith]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import json
import torch
import bench
from kid_amd.sharding import ShardedColumns

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
arith = sys.argv[2] if len(sys.argv) > 2 else "p64"
for name, ncol in (("config2", 10000), ("config3", 100000), ("config5", 100000)):
    st, iiwarm, desc = bench.make_workload(name, ncol)
    sh = ShardedColumns(st, 0, 1, 0, iiwarm, local=True, arith=arith)
    last = None
    bad = 0
    for s in range(0, steps, 50):
        for _ in range(50):
            sh.step(10.0)
        d = sh.diagnostics()
        precip = d["precip"].cpu().tolist(); sanity = d["sanity"].cpu().tolist()
        finite = all(torch.isfinite(v).all().item() for v in sh.st.values()) and torch.isfinite(sh.ppt).all().item()
        mono = last is None or all(a >= b for a, b in zip(precip, last))
        last = precip
        ok = finite and mono and sum(sanity[7:]) == 0 and sanity[0] < 0.05 and sanity[1] < 0.05 and sanity[3] < 0.05 and sanity[5] < 0.05
        bad += not ok
        print(json.dumps({"workload": name, "arith": arith, "steps": s + 50, "finite": finite, "precip_monotone": mono,
                          "negatives": int(sum(sanity[7:])), "max_qc_qr_nr_qs_qi_qg_ni": sanity[:7], "precip_sums": precip, "ok": ok}), flush=True)
    sh.close()
    print(json.dumps({"workload": name, "arith": arith, "intervals_not_ok": bad}), flush=True)
