#!/usr/bin/env python3
"""Where a wave's time goes between the two workgroup barriers of the column kernel (profiling build only:
make -C kid_amd/csrc prof; KIDMP_DEBUG_STOP=7 makes every wave leave shader-clock intervals in ppt).
usage: KIDMP_DEBUG_STOP=7 python tools/wave_timeline.py config3 [ncol]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from kid_amd import thompson

assert os.environ.get("KIDMP_DEBUG_STOP") in ("7", "8")
name = sys.argv[1]
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else (10000 if name == "config2" else 100000)
thompson.load_library(os.path.join(ROOT, "kid_amd", "libkidmp_prof.so"))
st, iiwarm, desc = bench.make_workload(name, ncol)
m = thompson.ThompsonMP(iiwarm=iiwarm)
dev = torch.device("cuda", 0)
d = {k: torch.as_tensor(v).contiguous().to(dev) for k, v in st.items()}
ppt = torch.zeros(ncol, 4, dtype=torch.float64, device=dev)
for _ in range(3):
    m.batch_step(d, 10.0, ppt)
torch.cuda.synchronize()
t = ppt.cpu().numpy()
names = ["pass 0", "barrier 1", "pass 1 (own bands)", "barrier 2"] if os.environ["KIDMP_DEBUG_STOP"] == "7" else ["pass 3", "pass 4", "barrier 3", "pass 5 (own bands)"]
print(desc)
for i, n in enumerate(names):
    x = t[:, i]
    print("%-20s mean %8.0f  median %8.0f  p90 %8.0f  max %8.0f cycles" % (n, x.mean(), np.median(x), np.percentile(x, 90), x.max()))
g = t[: ncol // 4 * 4].reshape(-1, 4, 4)
p1 = g[:, :, 2]
print("pass 1 per workgroup: mean of (max - min over its 4 waves) %.0f, mean of max %.0f, mean %.0f"
      % ((p1.max(1) - p1.min(1)).mean(), p1.max(1).mean(), p1.mean()))
tot = t.sum(1)
print("sum of the four intervals: mean %.0f cycles;  barrier share %.1f %%" % (tot.mean(), 100 * (t[:, 1] + t[:, 3]).mean() / tot.mean()))
