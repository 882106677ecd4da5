#!/usr/bin/env python3
"""The worst levels of one fuzz batch (tools/fuzz_campaign.py's generator), variable by variable: which output differs, by how
much, the oracle's own ulp-sensitivity there, the residue flags and the level's inputs.
    python tools/fuzz_level_diag.py <seed> <nz> <dt> [ncol=1000] [how many=6]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from kid_amd import ThompsonMP
from oracle.oracle import Oracle
from parity import FLOORS, OUT, branch_aware_compare, rel_err
from test_gpu_fuzz import fuzz_columns

seed, nz, dt = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
ncol = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
top = int(sys.argv[5]) if len(sys.argv) > 5 else 6
m, o = ThompsonMP(iiwarm=False), Oracle(iiwarm=False, nthreads=min(os.cpu_count() or 1, 16))
st = fuzz_columns(ncol, nz, seed)
got = {k: v.copy() for k, v in st.items()}
gppt, _ = m.batch_step_host(got, dt)
cmp = branch_aware_compare(o, st, dt, got, gppt, depletion=1e-5)
err, sens, flags, ref = cmp["err"], cmp["sens"], cmp["flags"], cmp["ref"]
for flat in np.argsort(-err.ravel())[:top]:
    c, k = np.unravel_index(flat, err.shape)
    per = {v: float(rel_err(got[v][c, k], ref[v][c, k], max(FLOORS[v], 1e-5 * abs(st[v][c, k])))) for v in OUT}
    w = max(per, key=per.get)
    print("col %d level %d: err %.2e (worst %s), oracle sensitivity %.1e, residue flags %d | T %.3f p %.0f in: qv %.3e qc %.2e qi %.2e qr %.2e "
          "qs %.2e qg %.2e ni %.2e nr %.2e | %s: hip %.10e oracle %.10e | all: %s"
          % (c, k, err[c, k], w, sens[c, k], flags[c, k], st["t"][c, k], st["p"][c, k], st["qv"][c, k], st["qc"][c, k], st["qi"][c, k],
             st["qr"][c, k], st["qs"][c, k], st["qg"][c, k], st["ni"][c, k], st["nr"][c, k], w, got[w][c, k], ref[w][c, k],
             {v: "%.1e" % e for v, e in per.items() if e > 1e-12}))
m.close(); o.close()
