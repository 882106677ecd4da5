#!/usr/bin/env python3
"""How many wave passes of the rate sweep (16 levels x 4 columns) hold only levels on which nothing can happen: no
hydrometeor above R1, not supersaturated over water, and (below freezing) less than 25 % supersaturated over ice.
Evolved states: the workload is stepped 10 times first.  usage: python tools/inactive_bands.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from kid_amd.sharding import ShardedColumns


def esat(T, ice):
    X = np.maximum(-80.0, T - 273.16)
    cw = [.611583699E03, .444606896E02, .143177157E01, .264224321E-1, .299291081E-3, .203154182E-5, .702620698E-8, .379534310E-11, -.321582393E-13]
    ci = [.609868993E03, .499320233E02, .184672631E01, .402737184E-1, .565392987E-3, .521693933E-5, .307839583E-7, .105785160E-9, .161444444E-12]
    c = ci if ice else cw
    e = np.zeros_like(X)
    for a in reversed(c):
        e = a + X * e
    return e


for name, ncol in (("config2", 10000), ("config3", 20000), ("config5", 20000)):
    st, iiwarm, desc = bench.make_workload(name, ncol)
    sh = ShardedColumns(st, 0, 1, 0, iiwarm, local=True)
    for nsteps in (0, 10, 100):
        while getattr(sh, "_n", 0) < nsteps:
            sh.step(10.0); sh._n = getattr(sh, "_n", 0) + 1
        s = {k: v.cpu().numpy() for k, v in sh.st.items()}
        T, p, qv = s["t"], s["p"], np.maximum(1e-10, s["qv"])
        es, ei = np.minimum(esat(T, False), 0.15 * p), np.minimum(esat(T, True), 0.15 * p)
        qvs, qvsi = 0.622 * es / (p - es), 0.622 * ei / (p - ei)
        qvsi = np.where(T <= 273.15, qvsi, qvs)
        ssatw, ssati = qv / qvs - 1, qv / qvsi - 1
        species = sum((s[k] > 1e-12) for k in ("qc", "qi", "qr", "qs", "qg")) > 0
        active = species | (ssatw > 1e-29) | ((T < 273.15) & (ssati >= 0.25))
        nz = T.shape[1]
        nb = (nz + 15) // 16
        g = active[: ncol // 4 * 4].reshape(-1, 4, nz)
        idle = sum(int((~g[:, :, b * 16:(b + 1) * 16].any(axis=(1, 2))).sum()) for b in range(nb))
        print("%s after %3d steps: inactive levels %.3f, wave passes with no active level %.3f of %d per workgroup"
              % (name, nsteps, 1 - active.mean(), idle / (g.shape[0] * nb), nb), flush=True)
    sh.close()
