#!/usr/bin/env python3
"""PCIe-inclusive rate of kidmp_batch_step_host against the pipeline's chunk size (page-locked host arrays).
usage: python tools/host_entry_sweep.py [config2|config3] [ncol]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from kid_amd import thompson
from kid_amd.thompson import FORCING_NAMES, STATE_NAMES

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else bench.DEFAULT_NCOL[name]
st, iiwarm, desc = bench.make_workload(name, ncol)
m = thompson.ThompsonMP(iiwarm=iiwarm)
keys = STATE_NAMES + FORCING_NAMES
h = {k: thompson.host_pinned_copy(np.ascontiguousarray(st[k])) for k in keys}
ppt = thompson.host_empty((ncol, 4)); ppt[...] = 0.0
print(desc)
for chunk in [0, 512, 1024, 2048, 4096, 8192, 16384, 32768, ncol]:
    if chunk > ncol:
        continue
    m.set_host_chunk(chunk)
    m.batch_step_host(h, 10.0, ppt=ppt)
    reps = 5 if ncol <= 20000 else 2
    t0 = time.perf_counter()
    for _ in range(reps):
        m.batch_step_host(h, 10.0, ppt=ppt)
    dt = (time.perf_counter() - t0) / reps
    print("chunk %6d: %8.3f ms per call  %.3e column-steps/s  in %.1f GB/s out %.1f GB/s"
          % (chunk, dt * 1e3, ncol / dt, 13472 * ncol / dt / 1e9, 11552 * ncol / dt / 1e9), flush=True)
