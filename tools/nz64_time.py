#!/usr/bin/env python
"""Kernel time of the warm-rain kernel on 20000 replicated warm columns resampled to nz = 64 (occupancy experiments:
at this size LDS does not limit the waves per SIMD, the VGPR budget does)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from kid_amd import ThompsonMP
st = cases.config2(1)
nz = 64
x0 = np.linspace(0, 1, 120); x1 = np.linspace(0, 1, nz)
col = {k: np.interp(x1, x0, v[0]) for k, v in st.items()}
col["dz"] = np.full(nz, 3000.0 / nz)
ncol = 20000
dev = {k: torch.from_numpy(np.ascontiguousarray(np.tile(v, (ncol, 1)))).cuda() for k, v in col.items()}
m = ThompsonMP(iiwarm=True)
ppt = torch.zeros(ncol, 4, dtype=torch.float64, device="cuda")
for _ in range(3): m.batch_step(dev, 10.0, ppt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): m.batch_step(dev, 10.0, ppt)
e1.record(); torch.cuda.synchronize()
print("nz=64 warm kernel ms per launch: %.4f" % (e0.elapsed_time(e1) / 20))
