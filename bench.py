#!/usr/bin/env python
"""bench.py -- Thompson mp column-steps/s (nz=120) on MI355X.

One "step" = one mp_thompson advance (dt = 10 s) of every column of the batch,
state resident in HBM.  Default workload = BASELINE.json configs[1]:
10^4 replicated warm-rain columns (the KiD 1-D warm case at t = 900 s), fp64,
per GPU (weak scaling: every rank owns its own 10^4 columns, no halo; RCCL only
for the final precipitation-diagnostics reduction).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     HBM roofline of the column-step kernel: achieved = 19232 B of
               algorithmic traffic per column-step x columns per
               launch / average launch duration (one HIP event pair on the
               launch stream around the K timed launches, / K), peak 8 TB/s.
  cpu_baseline the CPU oracle (a port of the reference, kind "port") timed on
               this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_FP64 = 19232          # (11 read + 9 written profiles) * 120 * 8 B + 32 B  
HBM_PEAK = 8.0e12
NZ = 120
DT = 10.0


def make_workload(name, ncol):
    """Returns (numpy state dict [ncol, nz], iiwarm, description).  Synthetic inputs only
    (tests/cases.py; the config-2 base column is the committed fixture tests/golden/)."""
    import cases
    if name == "config2":
        return (cases.config2(ncol), True,
                "config2: %d replicated warm-rain columns (KiD 1-D warm case at t=900 s)" % ncol)
    if name == "config3":
        return cases.config3(ncol), False, "config3: %d perturbed mixed-phase deep-convection columns" % ncol
    if name == "config5":
        return (cases.config5(ncol), False,
                "config5: %d sedimentation-heavy squall-line columns (>=20 CFL substeps)" % ncol)
    raise SystemExit("unknown workload " + name)


def measured_traffic(workload, ncol):
    """HBM bytes per launch of the column-step kernel from a committed rocprofv3 --pmc run of this
    same workload (profiles/rNN_pmc_<workload>.json, made by tools/pmc_profile.sh; FETCH_SIZE and
    WRITE_SIZE are in KiB and come from separate passes).  On gfx950 FETCH_SIZE reports half of the
    bytes of a coalesced stream; tools/calibrate_fetch.sh measured 1/1.84 for this kernel's 8-byte-
    per-lane loads on a known byte count, so the read side is doubled (MI355X_MICROARCH.md, HBM)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % workload)))
    if not found:
        return None
    d = json.load(open(found[-1]))                  # the latest round's profile
    if int(d.get("ncol", ncol)) != ncol or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None
    return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0


def cpu_baseline(model, st, iiwarm, budget_s=12.0):
    """The cpu_baseline leg: the oracle (a CPU port of the reference) on this host's cores on a
    bounded sample of the same workload, and -- with the same oracle output -- the accuracy figure
    of BASELINE.json: max relative |dq| of the HIP path after one step from identical inputs
    (conditioned levels only, see tests/parity.py)."""
    import numpy as np
    import torch
    from oracle.oracle import Oracle
    from parity import OUT, conditioned_mask, max_rel
    cores = os.cpu_count() or 1
    o = Oracle(iiwarm=iiwarm, nthreads=cores)
    nsamp = min(st["qv"].shape[0], 2000)
    s0 = {k: np.ascontiguousarray(v[:nsamp].copy()) for k, v in st.items()}

    # accuracy: one step from identical inputs
    nacc = min(nsamp, 256)
    sa = {k: np.ascontiguousarray(v[:nacc].copy()) for k, v in s0.items()}
    ref = {k: v.copy() for k, v in sa.items()}
    rppt = o.batch_step(ref, DT)
    mask = conditioned_mask(o, sa, DT, ref)
    dev = {k: torch.from_numpy(v).cuda() for k, v in sa.items()}
    ppt = torch.zeros(nacc, 4, dtype=torch.float64, device="cuda")
    model.batch_step(dev, DT, ppt)
    torch.cuda.synchronize()
    got = {k: dev[k].cpu().numpy() for k in OUT}
    mx, _ = max_rel(got, ref, OUT, mask)
    pm = float(np.max(np.abs(ppt.cpu().numpy() - rppt) / np.maximum(np.abs(rppt), 1e-12)))

    # timing: all cores, one column per task
    s = {k: v.copy() for k, v in s0.items()}
    o.batch_step(s, DT, nthreads=cores)           # page in
    done, t0 = 0, time.perf_counter()
    while True:
        o.batch_step(s, DT, nthreads=cores)
        done += nsamp
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    s1 = {k: np.ascontiguousarray(v[:200].copy()) for k, v in s0.items()}
    t1 = time.perf_counter()
    o.batch_step(s1, DT, nthreads=1)
    one = 200 / (time.perf_counter() - t1)
    o.close()
    return {"value": done / el, "unit": "column-steps/s", "cores": cores, "kind": "port",
            "sample": "%d columns of the same workload, %d steps, %.1f s, one column per task on %d threads"
                      % (nsamp, done // nsamp, el, cores),
            "single_core": one,
            "max_rel_dq_gpu_vs_cpu": max(mx, pm),
            "ill_conditioned_levels_excluded": [int((~mask).sum()), int(mask.size)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5"])
    ap.add_argument("--ncol", type=int, default=0, help="columns per GPU (default: the config's own size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank); gloo only to rehearse N ranks on fewer GPUs")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from kid_amd import STATE_NAMES, ThompsonMP

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > 1 and local >= ndev:
        raise SystemExit("rank %d needs its own GPU (found %d); use --backend gloo to rehearse" % (local, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))    # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"
    ncol = args.ncol or {"config2": 10000, "config3": 100000, "config5": 100000}[args.workload]

    st, iiwarm, desc = make_workload(args.workload, ncol)
    model = ThompsonMP(iiwarm=iiwarm, device=local)
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    ppt = torch.zeros(ncol, 4, dtype=torch.float64, device="cuda")
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        model.batch_step(dev, DT, ppt)
    sync_all()
    t0 = time.perf_counter()
    # One event pair around the K launches (an event per step costs a barrier packet per launch: -6 % throughput
    # on the 0.12 ms headline kernel).  torch's current stream == the stream batch_step launches on, and nothing
    # but the K column-step kernels runs between the two events.
    ev_begin.record()
    for i in range(args.steps):
        model.batch_step(dev, DT, ppt)
    ev_end.record()
    diag = model.reduce_ppt(ppt)           # domain sums of surface precipitation (W:248-275 analogue)
    if world > 1:
        diag = diag.to(coll_dev)
        dist.all_reduce(diag)              # the only collective: final diagnostics reduction
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms = ev_begin.elapsed_time(ev_end) / args.steps     # average launch duration (launch-to-launch, gaps included)

    if rank == 0:
        total_cols = ncol * world
        value = total_cols * args.steps / elapsed
        achieved = ALGO_BYTES_FP64 * ncol / (kern_ms * 1e-3)
        tbytes = measured_traffic(args.workload, ncol)
        out = {
            "metric": "thompson_mp_column_steps_per_sec", "value": value, "unit": "column-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + ", nz=120, dt=10 s, fp64", "ncol_per_gpu": ncol, "nz": NZ, "dt": DT,
                       "parallelism": "columns sharded over ranks, no halo; one RCCL all-reduce of 4 precipitation sums"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK,
                         "traffic": None if tbytes is None else tbytes / (kern_ms * 1e-3) / 1e9,
                         "traffic_bytes_per_launch": tbytes,
                         "kernel": ThompsonMP.kernel_name(), "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_column_step": ALGO_BYTES_FP64,
                         "note": "fp64 transcendental-bound path (SURVEY 8d): the HBM fraction is reported as "
                                 "mandated; see DESIGN.md for the VALU-side accounting"},
            "precip_domain_sums": [float(x) for x in diag.cpu().tolist()],
            "init_seconds": model.init_seconds,
        }
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the host cores are shared by the ranks)
            out["cpu_baseline"] = cpu_baseline(model, st, iiwarm)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
