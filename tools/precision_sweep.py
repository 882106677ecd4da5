#!/usr/bin/env python
"""BASELINE config 5: "sedimentation-heavy squall-line profile, >= 20 CFL substeps, 10^5 columns, fp32 vs fp64
tolerance sweep".  Three arithmetics of the same kernel source, from identical (binary32-representable) inputs:

  p64    every REAL and DOUBLE PRECISION of the reference in binary64 (the parity build)
  p32n   the reference AS SHIPPED: what it declares REAL in binary32, its DOUBLE PRECISION rates in binary64
  f32    everything binary32

Per arithmetic and step count (1, 6, 60): percentiles of |x - x_p64| / max(|x_p64|, floor) per variable over all
levels (floors 1e-8 kg/kg, 1e-2 kg^-1, 1 K), relative difference of the domain precipitation sums; the substep
histogram that defines the workload (identical in all three unless a level sits at a CFL threshold), and the kernel
time per launch of each arithmetic (HIP events around 10 launches).  One JSON line each.

    python tools/precision_sweep.py [--ncol 100000] > profiles/rNN_precision_sweep_config5.jsonl
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from kid_amd import STATE_NAMES, ThompsonMP  # noqa: E402
from parity import FLOORS  # noqa: E402

ALGO_BYTES = {"p64": 19232, "p32n": 9616, "f32": 9616}        # fp64 19 232 B, fp32 9 616 B per column-step


def run(model, st32, arith, nsteps, checkpoints):
    dt = torch.float64 if arith == "p64" else torch.float32
    dev = {k: torch.from_numpy(v).cuda().to(dt) for k, v in st32.items()}
    ncol = st32["qv"].shape[0]
    ppt = torch.zeros(ncol, 4, dtype=dt, device="cuda")
    nstep = torch.zeros(ncol, 4, dtype=torch.int32, device="cuda")
    out, hist = {}, None
    for n in range(1, nsteps + 1):
        if arith == "p64":
            model.batch_step(dev, 10.0, ppt, nstep=nstep)
        else:
            model.batch_step32(dev, 10.0, ppt, arith=arith, nstep=nstep)
        if n == 1:
            ns = nstep.cpu().numpy()
            hist = {sp: {int(v): int(c) for v, c in zip(*np.unique(ns[:, i], return_counts=True))}
                    for i, sp in enumerate(("rain", "ice", "snow", "graupel"))}
        if n in checkpoints:
            torch.cuda.synchronize()
            out[n] = ({k: dev[k].double().cpu().numpy() for k in STATE_NAMES}, ppt.double().sum(dim=0).cpu().numpy())
    # kernel time: 10 more launches between two events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        if arith == "p64":
            model.batch_step(dev, 10.0, ppt)
        else:
            model.batch_step32(dev, 10.0, ppt, arith=arith)
    e1.record()
    torch.cuda.synchronize()
    return out, hist, e0.elapsed_time(e1) / 10.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config5", choices=["config3", "config5"])
    ap.add_argument("--ncol", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    st32 = {k: np.ascontiguousarray(v.astype(np.float32)) for k, v in getattr(cases, args.workload)(args.ncol).items()}
    model = ThompsonMP(iiwarm=False)
    cps = sorted({1, 6, args.steps})
    res = {a: run(model, st32, a, args.steps, cps) for a in ("p64", "p32n", "f32")}
    ref = res["p64"][0]
    print(json.dumps({"workload": args.workload, "columns": args.ncol, "substep_histogram_first_step": res["p64"][1],
                      "substep_counts_equal_in_all_arithmetics": res["p64"][1] == res["p32n"][1] == res["f32"][1]}))
    for a in ("p64", "p32n", "f32"):
        ms = res[a][2]
        print(json.dumps({"arithmetic": a, "kernel_ms_per_launch": ms, "column_steps_per_s": args.ncol / (ms * 1e-3),
                          "algorithmic_bytes_per_column_step": ALGO_BYTES[a],
                          "hbm_roofline_frac": ALGO_BYTES[a] * args.ncol / (ms * 1e-3) / 8.0e12}))
    for a in ("p32n", "f32"):
        for n in cps:
            got, gp = res[a][0][n]
            r, rp = ref[n]
            per = {}
            for k in ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr", "t"):
                e = np.abs(got[k] - r[k]) / np.maximum(np.abs(r[k]), 1e4 * FLOORS[k] if k != "t" else 1.0)
                per[k] = {"median": float(np.median(e)), "q99": float(np.quantile(e, 0.99)),
                          "q99.9": float(np.quantile(e, 0.999)), "max": float(e.max())}
            print(json.dumps({"arithmetic": a, "vs": "p64", "steps": n, "rel_diff_percentiles": per,
                              "precip_sum_rel_diff": [float(x) for x in np.abs(gp - rp) / np.maximum(np.abs(rp), 1e-8)]}))
    model.close()


if __name__ == "__main__":
    main()
