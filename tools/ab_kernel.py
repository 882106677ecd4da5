#!/usr/bin/env python3
"""A/B timing of several builds of libkidmp.so in ONE process and ONE GPU session (box-to-box noise is ~3 %, launch-to-
launch noise within a session ~0.5 %): every library gets its own context, the same device-resident columns, and the
builds are timed alternately.  Also reports whether the builds' results after one step are bit-identical to the first
library's (most optimisations are meant to be), else the largest relative difference per variable.

usage: ab_kernel.py libA.so libB.so [...] [--workloads config3 config5 config2] [--steps 20] [--reps 5] [--ncol N]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
from kid_amd.thompson import _Cfg, STATE_NAMES

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workloads", nargs="+", default=["config3", "config5", "config2"])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--ncol", type=int, default=0)
ap.add_argument("--fresh", action="store_true",
                help="every timed launch starts from the workload's initial state (restored outside the event pair): for "
                     "truncated builds (tools/pass1_blocks.sh), whose state does not evolve, so that all builds do the same work")
args = ap.parse_args()
vp = C.c_void_p


def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.kidmp_init.restype = C.c_int
    L.kidmp_init.argtypes = [C.POINTER(_Cfg), C.POINTER(vp)]
    L.kidmp_finalize.argtypes = [vp]
    L.kidmp_last_error.restype = C.c_char_p
    L.kidmp_last_error.argtypes = [vp]
    L.kidmp_batch_step_device.restype = C.c_int
    L.kidmp_batch_step_device.argtypes = [vp, C.c_int64, C.c_int32, C.c_double] + [vp] * 18 + [vp]
    return L


libs = [load(p) for p in args.libs]
dev = torch.device("cuda", 0)
for w in args.workloads:
    ncol = args.ncol or bench.DEFAULT_NCOL[w]
    st, iiwarm, desc = bench.make_workload(w, ncol)
    ctxs = []
    for L in libs:
        cfg = _Cfg(int(iiwarm), 1, 100.0, 0, 0)
        h = vp()
        rc = L.kidmp_init(C.byref(cfg), C.byref(h))
        assert rc == 0, L.kidmp_last_error(None)
        ctxs.append(h)

    def fresh():
        return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in st.items()}

    def step(i, d, ppt):
        s = torch.cuda.current_stream().cuda_stream
        a = [d[k].data_ptr() for k in STATE_NAMES] + [d["p"].data_ptr(), None, d["dz"].data_ptr(), ppt.data_ptr(), None, None]
        rc = libs[i].kidmp_batch_step_device(ctxs[i], ncol, d["qv"].shape[1], 10.0, *a, s)
        assert rc == 0, libs[i].kidmp_last_error(ctxs[i])

    # results after one step from identical inputs
    outs = []
    for i in range(len(libs)):
        d, ppt = fresh(), torch.zeros(ncol, 4, dtype=torch.float64, device=dev)
        step(i, d, ppt)
        torch.cuda.synchronize()
        outs.append(({k: d[k].cpu().numpy() for k in STATE_NAMES}, ppt.cpu().numpy()))
    for i in range(1, len(libs)):
        same = all(np.array_equal(outs[i][0][k], outs[0][0][k]) for k in STATE_NAMES) and np.array_equal(outs[i][1], outs[0][1])
        if same:
            print("%s: %s == %s bit for bit after one step" % (w, os.path.basename(args.libs[i]), os.path.basename(args.libs[0])))
        else:
            worst = {}
            for k in STATE_NAMES:
                a, b = outs[i][0][k], outs[0][0][k]
                fl = 1.0 if k == "t" else (1e-6 if k.startswith("n") else 1e-12)
                worst[k] = float((np.abs(a - b) / np.maximum(np.abs(b), fl)).max())
            nd = int(sum((outs[i][0][k] != outs[0][0][k]).any(axis=1).sum() for k in ("qv", "qc", "qr", "qi", "qs", "qg", "ni", "nr", "t")))
            print("%s: %s differs from %s: max rel per variable %s; ppt %.2e; column-variables touched %d" % (
                w, os.path.basename(args.libs[i]), os.path.basename(args.libs[0]),
                {k: "%.1e" % v for k, v in worst.items() if v > 0}, float(np.abs(outs[i][1] - outs[0][1]).max()), nd))
    # timing: evolving state per library (each advances its own copy), alternating
    states = [(fresh(), torch.zeros(ncol, 4, dtype=torch.float64, device=dev)) for _ in libs]
    for i in range(len(libs)):
        for _ in range(3):
            step(i, *states[i])
    torch.cuda.synchronize()
    times = [[] for _ in libs]
    pristine = fresh() if args.fresh else None
    for rep in range(args.reps):
        for i in range(len(libs)):
            if args.fresh:
                tot = 0.0
                for _ in range(args.steps):
                    for k in pristine:
                        states[i][0][k].copy_(pristine[k])
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    step(i, *states[i])
                    e1.record()
                    torch.cuda.synchronize()
                    tot += e0.elapsed_time(e1)
                times[i].append(tot / args.steps)
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step(i, *states[i])
            e1.record()
            torch.cuda.synchronize()
            times[i].append(e0.elapsed_time(e1) / args.steps)
    for i, p in enumerate(args.libs):
        t = sorted(times[i])
        print("%s %-28s median %.4f ms  min %.4f  max %.4f   (vs first: %+.2f %%)" % (
            w, os.path.basename(p), t[len(t) // 2], t[0], t[-1], 100.0 * (t[len(t) // 2] / sorted(times[0])[len(times[0]) // 2] - 1)))
    for L, h in zip(libs, ctxs):
        L.kidmp_finalize(h)
    del states
    torch.cuda.empty_cache()
