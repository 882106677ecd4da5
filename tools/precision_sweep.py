#!/usr/bin/env python
"""BASELINE config 5 "fp32 vs fp64 tolerance sweep" (round-1 form): the fp64 kernel run with the prognostic
state rounded to a narrower storage format after every step, against the pure fp64 run.

  f64      state kept in binary64 (the parity build, P64)
  f32      state rounded to binary32 after every step: what KiD's default REAL storage does to the arrays
           between calls (the reference's native "P32n" build additionally keeps its local REALs in fp32)
  bf16x2   (for scale) state rounded to 16 significand bits

Prints, per storage format and step count, max relative difference vs f64 over the conditioned levels
(floors as tests/parity.py) and the relative difference of the domain precipitation sums."""
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


def round_state(dev, mode):
    if mode == "f64":
        return
    for k in STATE_NAMES:
        x = dev[k]
        if mode == "f32":
            x.copy_(x.float().double())
        else:                                   # keep 16 significand bits
            m, e = torch.frexp(x)
            x.copy_(torch.ldexp(torch.round(m * 65536.0) / 65536.0, e))


def run(model, st, mode, nsteps, checkpoints):
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    round_state(dev, mode)
    ppt = torch.zeros(st["qv"].shape[0], 4, dtype=torch.float64, device="cuda")
    out = {}
    for n in range(1, nsteps + 1):
        model.batch_step(dev, 10.0, ppt)
        round_state(dev, mode)
        if n in checkpoints:
            torch.cuda.synchronize()
            out[n] = ({k: dev[k].cpu().numpy() for k in STATE_NAMES}, ppt.sum(dim=0).cpu().numpy())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config5", choices=["config3", "config5"])
    ap.add_argument("--ncol", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    st = getattr(cases, args.workload)(args.ncol)
    m = ThompsonMP(iiwarm=False)
    cps = sorted({1, 10, args.steps})
    # CFL substep counts of the first three steps (rain, ice, snow, graupel): the workload is defined by them
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    ppt0 = torch.zeros(args.ncol, 4, dtype=torch.float64, device="cuda")
    nst = torch.zeros(args.ncol, 4, dtype=torch.int32, device="cuda")
    for n in (1, 2, 3):
        m.batch_step(dev, 10.0, ppt0, nstep=nst)
        h = nst.cpu().numpy()
        print(json.dumps({"nstep_histogram_step": n, **{name: {int(v): int(c) for v, c in zip(*np.unique(h[:, i], return_counts=True))}
                                                         for i, name in enumerate(("rain", "ice", "snow", "graupel"))}}))
    ref = run(m, st, "f64", args.steps, cps)
    rows = []
    for mode in ("f32", "bf16x2"):
        got = run(m, st, mode, args.steps, cps)
        for n in cps:
            per = {}
            for k in ("qv", "qc", "qr", "qi", "qs", "qg", "ni", "nr", "t"):
                e = np.abs(got[n][0][k] - ref[n][0][k]) / np.maximum(np.abs(ref[n][0][k]), FLOORS[k])
                per[k] = [float(np.quantile(e, 0.999)), float(e.max())]
            p = np.abs(got[n][1] - ref[n][1]) / np.maximum(np.abs(ref[n][1]), 1e-300)
            rows.append({"storage": mode, "steps": n, "q999_and_max_rel_diff": per, "precip_sum_rel_diff": [float(x) for x in p]})
            print(json.dumps(rows[-1]))
    return rows


if __name__ == "__main__":
    main()
