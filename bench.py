#!/usr/bin/env python
"""bench.py -- Thompson mp column-steps/s (nz=120) on MI355X.

One "step" = one mp_thompson advance (dt = 10 s) of every column of the batch, state resident in HBM.
Headline workload = BASELINE.json configs[2]: 10^5 perturbed mixed-phase deep-convection columns (ice, snow and
graupel active), fp64, per GPU -- the largest single-GPU configuration (BASELINE.json's metric names no config).
Weak scaling: every rank owns its own columns, no halo, no data-path collective; RCCL only for the final diagnostics
reduction: one all-gather of the per-rank exact precipitation accumulators (the domain means of W:248-303; 24 int64 limbs,
summed locally: the same digits for every rank count), inside the timed region; the optional max-q / negative-value scan of the end state runs after the clock and only
feeds the printed line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|config4|config5]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: it starts N fresh
rank processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, 127.0.0.1 rendezvous) before anything touches the
GPU, rank r binds GPU r, and rank 0 prints the line.  Under `python -m torch.distributed.run` the ranks already
exist and the script is simply rank RANK.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline         HBM roofline of the column-step kernel: achieved = 19232 B of algorithmic traffic per
                   column-step x columns per launch / average launch duration (one HIP event pair on
                   the launch stream around the K timed launches, / K), peak 8 TB/s; `valu_frac` = the fp64-VALU
                   issue fraction that actually binds (SURVEY 8d asks for both), `valu_busy_frac` the same with the
                   measured issue time, `valu_useful_frac` = busy x active-lane fraction; `valu_floor_ms` = the
                   kernel time at perfect VALU issue of the present instruction count (what the 40 % HBM target would
                   need is stated in `note`); `traffic` from the committed rocprofv3 --pmc profile of the SAME code
                   object (fingerprint-checked, else null).
  cpu_baseline     the CPU oracle (a C port of the reference, kind "port") timed on this host's cores (persistent
                   thread pool, >= 512 columns per thread, thread count chosen from a short ladder and printed) on a
                   bounded sample of the same workload.
  accuracy         max relative |dq| of the HIP path vs the oracle after one step from identical inputs over >= 2048
                   columns: `max_rel_dq` is the maximum over ALL levels (nothing filtered), with the number of levels
                   beyond 1e-10 and the fraction of columns entirely within 1e-10 beside it;
                   `max_rel_dq_steady_levels_only` is the same maximum restricted to levels where the oracle itself
                   moves by <= 1e-11 under 2-4 ulp input perturbations (a FILTERED figure, named as such).
  other_workloads  the other single-GPU configs of BASELINE.json (config 2 and 5 at N = 1; config 4 =
                   125 000 mixed-phase columns per rank at N > 1), timed in the same process after the headline,
                   each with its own full roofline / cpu_baseline / accuracy objects.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_FP64 = 19232          # (11 read + 9 written profiles) * 120 * 8 B + 32 B  
ALGO_BYTES_WARM = 11528          # warm-only variant: 12 profiles * 120 * 8 B + 8 B   (quoted alongside)
HBM_PEAK = 8.0e12
N_SIMD = 256 * 4                 # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                 # peak engine clock (MI355X_MICROARCH.md)
VALU_CYCLES_PER_INSTR = 4        # one wave64 VALU instruction occupies its SIMD for 4 cycles (fp64 FMA/MUL/ADD: full rate)
NZ = 120
DT = 10.0
DEFAULT_NCOL = {"config2": 10000, "config3": 100000, "config4": 125000, "config5": 100000}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config3", choices=sorted(DEFAULT_NCOL))
    ap.add_argument("--ncol", type=int, default=0, help="columns per GPU (default: the config's own size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="time the named workload only")
    ap.add_argument("--no-host-entry", action="store_true", help="skip the PCIe-inclusive leg (kidmp_batch_step_host)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank); gloo only to rehearse N ranks on fewer GPUs")
    ap.add_argument("--arith", default="p64", choices=["p64", "p32n", "f32"],
                    help="p64 = the parity build (default, the metric's dtype); p32n = the reference as shipped (binary32 state, "
                         "binary64 rates); f32 = all binary32 (BASELINE config 5's precision sweep)")
    ap.add_argument("--lib", default=None, help="an explicitly named build of libkidmp.so (A/B and profiling builds)")
    ap.add_argument("--rehearse-launcher", action="store_true",
                    help="CPU-only boxes: exercise launcher + rendezvous + reduction with NO physics (value is 0)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------ launcher (never touches the GPU)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """Start n rank processes of this script and wait for them.  The parent imports neither torch nor the HIP
    library, so no GPU context exists in it; children are started as ordinary subprocesses (no exec from a
    GPU-initialised process)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:                      # one rank failed: stop exactly the ranks we started
                    q.terminate()
    return rc


# ------------------------------------------------------------------ workloads
def make_workload(name, ncol, rank=0):
    """Returns (numpy state dict [ncol, nz], iiwarm, description).  Synthetic inputs only (tests/cases.py; the
    config-2 base column is the committed fixture tests/golden/).  Perturbed workloads use rank-offset seeds so
    that the ranks of a multi-GPU run hold different columns."""
    import cases
    if name == "config2":
        return (cases.config2(ncol), True,
                "config2: %d replicated warm-rain columns (KiD 1-D warm case at t=900 s)" % ncol)
    if name == "config3":
        return (cases.config3(ncol, seed=cases.SEED + rank), False,
                "config3: %d perturbed mixed-phase deep-convection columns" % ncol)
    if name == "config4":
        return (cases.config3(ncol, seed=cases.SEED + rank), False,
                "config4: %d perturbed mixed-phase columns per GPU (10^6 sharded over 8 GPUs at N=8)" % ncol)
    if name == "config5":
        return (cases.config5(ncol, seed=cases.SEED + rank), False,
                "config5: %d sedimentation-heavy squall-line columns (>=20 CFL substeps)" % ncol)
    raise SystemExit("unknown workload " + name)


def load_pmc_profile(workload, ncol, fingerprint, arith="p64"):
    """The committed rocprofv3 --pmc run of this workload and arithmetic (profiles/rNN_pmc_<workload>[_<arith>].json,
    made by tools/pmc_profile.sh; counters come from separate passes), or None when it was measured on another build
    of the kernel: the profile carries the code object's fingerprint (kidmp_kernel_fingerprint) and must match."""
    import glob
    w = "config3" if workload == "config4" else workload          # config 4 is config 3's recipe
    if arith != "p64":
        w += "_" + arith
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % w)))
    if not found:
        return None
    d = json.load(open(found[-1]))                  # the latest round's profile
    if d.get("fingerprint") != fingerprint:
        return None
    d["_file"] = os.path.relpath(found[-1], ROOT)
    return d


def host_entry_leg(model, st, iiwarm, reps=3):
    """The PCIe-inclusive rate of the drop-in boundary (kidmp_batch_step_host: the model's arrays live on the host,
    as in KiD): whole calls timed by the host clock.  `pinned` / `pageable`: all 14 + 12 profiles cross PCIe, from
    page-locked memory (what the Fortran shim stages through) or ordinary numpy arrays; `kid_adapter`: the call the
    KiD adapter makes -- the arrays KiD never fills (nc, nwfa, nifa; in a warm run the frozen species) are left out
    and formed on the GPU.  Reported beside the headline, never as `value`."""
    import time
    import numpy as np
    from kid_amd import thompson
    from kid_amd.thompson import FORCING_NAMES, STATE_NAMES
    keys = [k for k in STATE_NAMES + FORCING_NAMES if k != "w"]
    left_out = ("nc", "nwfa", "nifa") + (("qi", "qs", "qg", "ni") if iiwarm else ())
    ncol, nz = st["qv"].shape
    out = {"unit": "column-steps/s", "ncol": ncol,
           "note": "kidmp_batch_step_host, host clock around whole calls (upload + step + download pipelined over column "
                   "chunks); w is not uploaded for a non-aerosol context"}
    for kind in ("pinned", "kid_adapter", "pageable"):
        use = [k for k in keys if not (kind == "kid_adapter" and k in left_out)]
        if kind == "pageable":
            h = {k: np.ascontiguousarray(st[k]).copy() for k in use}
            ppt = np.zeros((ncol, 4))
        else:
            h = {k: thompson.host_pinned_copy(np.ascontiguousarray(st[k])) for k in use}
            ppt = thompson.host_empty((ncol, 4)); ppt[...] = 0.0
        n_in, n_out = len(use), len([k for k in use if k not in ("p", "dz")])
        model.batch_step_host(h, DT, ppt=ppt)                # warm-up: staging ring allocated, pages touched
        t0 = time.perf_counter()
        for _ in range(reps):
            model.batch_step_host(h, DT, ppt=ppt)
        dt = (time.perf_counter() - t0) / reps
        b_in, b_out = n_in * nz * 8 + 32, n_out * nz * 8 + 32
        out[kind] = {"value": ncol / dt, "ms_per_call": dt * 1e3, "profiles_in_out": [n_in, n_out],
                     "GBps_each_way": [b_in * ncol / dt / 1e9, b_out * ncol / dt / 1e9]}
        del h, ppt
    return out


def host_cores():
    """(threads this process may run at once, how that was found): the scheduler affinity mask, cut to the cgroup CPU
    quota when there is one (a GPU box hands each job a share of the host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    how = "sched_getaffinity"
    try:                                                     # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(q) > 0:
            lim = max(1, int(int(q) / int(per)))
            if lim < n:
                n, how = lim, "cgroup cpu.max"
    except (OSError, ValueError):
        try:                                                 # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and max(1, q // per) < n:
                n, how = max(1, q // per), "cgroup cfs_quota"
        except (OSError, ValueError):
            pass
    return n, how


ACCURACY_COLUMNS = {"config2": 2048, "config3": 20000, "config4": 20000, "config5": 20000}   # replicas need no more; the
# perturbed workloads show their accuracy tail (conditioning of the saturation adjustment) only on samples of this size


def accuracy_leg(model, st, iiwarm, nacc=2048):
    """The accuracy figure of BASELINE.json: max relative |dq| of the HIP path against the oracle after one step from
    identical inputs, over `nacc` columns.  Levels that sit on the reference's two chaotic `> 0.` tests (M:3587,
    M:3596) are compared against BOTH admissible outcomes (tests/parity.py).  `max_rel_dq` is the maximum over every
    level; the filtered maximum (levels the oracle itself holds steady under ulp perturbations) is reported under a
    name that says so."""
    import numpy as np
    import torch
    from oracle.oracle import Oracle
    from parity import OUT, TOL, branch_aware_compare, verdict
    cores, _ = host_cores()
    o = Oracle(iiwarm=iiwarm, nthreads=min(cores, 64))
    nacc = min(st["qv"].shape[0], nacc)
    sa = {k: np.ascontiguousarray(v[:nacc].copy()) for k, v in st.items()}
    dev = {k: torch.from_numpy(v).cuda() for k, v in sa.items()}
    ppt = torch.zeros(nacc, 4, dtype=torch.float64, device="cuda")
    model.batch_step(dev, DT, ppt)
    torch.cuda.synchronize()
    got = {k: dev[k].cpu().numpy() for k in OUT}
    res = verdict(branch_aware_compare(o, sa, DT, got, ppt.cpu().numpy()), tol=TOL)
    o.close()
    return {"max_rel_dq": max(res["max_err_any"], res.get("max_rel_ppt", 0.0)), "tolerance": TOL,
            "levels_beyond_1e-10": res["n_beyond_tol"], "cols_within_1e-10_frac": res["cols_within_tol_frac"],
            "max_rel_dq_steady_levels_only": res["max_rel"],
            "steady_means": "levels where the oracle's own output moves by <= 1e-11 under 2-4 ulp perturbations of T and q",
            "levels_not_steady": res["n_sensitive"], "oracle_max_sensitivity": res["max_sens"],
            "max_rel_precip": res.get("max_rel_ppt"), "levels": res["n_levels"], "columns": nacc,
            "levels_on_chaotic_branches": res["n_branch_levels"],
            "levels_matching_neither_branch": res["n_unmatched"],
            "note": "HIP vs the CPU oracle (a C restatement pinned on the survey's 6-7-digit reference outputs only: "
                    "parity against the reference itself is unpinned beyond those digits)"}


def cpu_baseline(st, iiwarm, budget_s=12.0):
    """The cpu_baseline leg: the oracle (a CPU port of the reference) on this host's cores on a bounded sample of the
    same workload.  The oracle's batch entry runs on a persistent thread pool (no thread creation inside the timed
    region), >= 512 columns per thread and call; the thread count is the best of a short ladder up to the cores this
    process may use, and is printed."""
    import numpy as np
    from oracle.oracle import Oracle
    avail, how = host_cores()
    ncol = st["qv"].shape[0]
    o = Oracle(iiwarm=iiwarm, nthreads=1)
    s1 = {k: np.ascontiguousarray(v[:512].copy()) for k, v in st.items()}
    o.batch_step(s1, DT, nthreads=1)                              # page in
    n1 = s1["qv"].shape[0]
    t1 = time.perf_counter()
    o.batch_step(s1, DT, nthreads=1)
    single = n1 / (time.perf_counter() - t1)
    ladder = {}
    best_n, best_v = 1, single
    for n in sorted({min(c, avail) for c in (8, 16, 32, 64, 128, 256, avail)}):
        nsamp = min(ncol, max(2048, 512 * n))
        s = {k: np.ascontiguousarray(v[:nsamp].copy()) for k, v in st.items()}
        o.batch_step(s, DT, nthreads=n)                           # threads created, pages touched
        t0 = time.perf_counter()
        o.batch_step(s, DT, nthreads=n)
        v = nsamp / (time.perf_counter() - t0)
        ladder[str(n)] = v
        if v > best_v * 1.03:                                     # more threads only if they pay
            best_n, best_v = n, v
    cores = best_n
    nsamp = min(ncol, max(2048, 512 * cores))
    s = {k: np.ascontiguousarray(v[:nsamp].copy()) for k, v in st.items()}
    done, t0 = 0, time.perf_counter()
    while True:
        o.batch_step(s, DT, nthreads=cores)
        done += nsamp
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    # one core on the same (evolved) columns the pool has just stepped: the yardstick of parallel_efficiency
    s1 = {k: np.ascontiguousarray(v[:512].copy()) for k, v in s.items()}
    t1 = time.perf_counter()
    o.batch_step(s1, DT, nthreads=1)
    single = s1["qv"].shape[0] / (time.perf_counter() - t1)
    o.close()
    value = done / el
    return {"value": value, "unit": "column-steps/s", "cores": cores, "kind": "port",
            "sample": "%d columns of the same workload, %d steps, %.1f s of wall clock, %d threads of a persistent pool "
                      "(%d columns per thread and call)" % (nsamp, done // nsamp, el, cores, nsamp // cores),
            "single_core": single, "parallel_efficiency": value / (single * cores),
            "host_cpu_count": os.cpu_count(), "cores_available": avail, "cores_available_from": how,
            "thread_ladder": ladder,
            "note": "C port of the reference (oracle/), about 20 % slower per core than the reference's own Fortran "
                    "measured in the survey (1.8e4 column-steps/s/core, warm, another CPU); a reported baseline, not a target"}


class _RehearsalShard:
    """--rehearse-launcher only (CPU box, no HIP device): stands in for ShardedColumns so that the launcher, the
    rendezvous and the reduction plumbing can be exercised by the CPU test-suite.  It runs NO physics; the line it
    produces reports value 0 and says so."""

    def __init__(self, ncol, rank):
        import torch
        self.ncol, self.rank = ncol, rank
        self.ppt = torch.zeros(ncol, 4, dtype=torch.float64)

    def step(self, dt):
        self.ppt[:, 0] += 1e-3 * dt * (self.rank + 1)

    def synchronize(self):
        pass

    def diagnostics(self, cpu_collective=True, want_sanity=True):
        import torch
        import torch.distributed as dist
        precip = self.ppt.sum(dim=0)
        sanity = torch.zeros(15, dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(precip)
        return dict(precip=precip, sanity=sanity, rates=None)


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus not in (1, world) and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; running %d ranks" % (args.gpus, world, world), file=sys.stderr)
    rehearse = args.rehearse_launcher
    if rehearse:
        if args.backend != "gloo" or torch.cuda.is_available():
            raise SystemExit("--rehearse-launcher is for CPU-only boxes with --backend gloo; on a GPU box run the real thing")
    else:
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
    cpu_coll = args.backend != "nccl"
    coll_dev = "cpu" if cpu_coll else "cuda"

    def sync_all():
        if not rehearse:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not rehearse:
                torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if not rehearse:
        import kid_amd
        from kid_amd.sharding import ShardedColumns
        kid_amd.load_library(args.lib)

    def time_workload(name, ncol, steps, warmup):
        """W untimed + K timed steps of one workload on this rank's shard; returns (shard, state, result dict)."""
        if rehearse:
            st, iiwarm, desc = None, True, "launcher rehearsal: %d placeholder columns, NO physics" % ncol
            shard = _RehearsalShard(ncol, rank)
        else:
            st, iiwarm, desc = make_workload(name, ncol, rank)
            shard = ShardedColumns(st, rank, world, local, iiwarm, local=True, arith=args.arith)
        ev0 = ev1 = None
        if not rehearse:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(warmup):
            shard.step(DT)
        if warmup > 0:
            shard.diagnostics(cpu_collective=cpu_coll)        # warm-up includes the collective (lazy channel set-up)
        sync_all()
        t0 = time.perf_counter()
        # One event pair around the K launches (an event per step costs a barrier packet per launch: -6 % on the
        # headline kernel).  torch's current stream == the stream the steps are launched on, and nothing but the
        # K column-step kernels runs between the two events.
        if ev0:
            ev0.record()
        for _ in range(steps):
            shard.step(DT)
        if ev1:
            ev1.record()
        diag = shard.diagnostics(cpu_collective=cpu_coll, want_sanity=False)   # the exchange of the path: W:248-303's precipitation sums
        sync_all()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        # after the clock: the optional max-q / negative-value scan of the end state, for the printed line only
        diag = dict(diag, sanity=shard.diagnostics(cpu_collective=cpu_coll)["sanity"])
        kern_ms = ev0.elapsed_time(ev1) / steps if ev0 else 0.0   # average launch duration (launch-to-launch)
        res = {"workload": desc + ", nz=120, dt=10 s, " + {"p64": "fp64", "p32n": "P32n (binary32 state, binary64 rates)",
                                                            "f32": "fp32"}[args.arith], "name": name, "ncol_per_gpu": ncol, "iiwarm": iiwarm,
               "value": 0.0 if rehearse else ncol * world * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
               "kernel_ms": kern_ms, "precip_domain_sums": [float(x) for x in diag["precip"].cpu().tolist()],
               "sanity_max_qc_qr_nr_qs_qi_qg_ni": [float(x) for x in diag["sanity"][:7].cpu().tolist()],
               "negative_values": int(diag["sanity"][7:].sum().item())}
        return shard, st, res

    def roofline(shard, res):
        ncol, kern_s = res["ncol_per_gpu"], res["kernel_ms"] * 1e-3
        algo_bytes = ALGO_BYTES_FP64 if args.arith == "p64" else ALGO_BYTES_FP64 // 2      # fp32 9 616 B
        achieved = algo_bytes * ncol / kern_s
        fp = shard.model.kernel_fingerprint(args.arith)
        prof = load_pmc_profile(res["name"], ncol, fp, args.arith)
        out = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
               "frac": achieved / HBM_PEAK, "traffic": None, "traffic_bytes_per_launch": None, "valu_frac": None,
               "valu_busy_frac": None,
               "kernel": shard.model.kernel_name(), "kernel_ms": res["kernel_ms"], "kernel_fingerprint": fp,
               "algorithmic_bytes_per_column_step": algo_bytes}
        if res["iiwarm"] and args.arith == "p64":
            out["frac_warm_only_bytes"] = ALGO_BYTES_WARM * ncol / kern_s / HBM_PEAK
            out["algorithmic_bytes_warm_only"] = ALGO_BYTES_WARM
        if prof is not None and "FETCH_SIZE" in prof and "WRITE_SIZE" in prof:
            # FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced stream
            # (tools/calibrate_fetch.sh measured 1/1.84 for this kernel's 8-byte-per-lane loads): read side doubled.
            per_col = (2.0 * prof["FETCH_SIZE"] + prof["WRITE_SIZE"]) * 1024.0 / float(prof["ncol"])
            out["traffic_bytes_per_launch"] = per_col * ncol
            out["traffic"] = per_col * ncol / kern_s / 1e9
            out["traffic_over_algorithmic"] = per_col / algo_bytes
            out["profile"] = prof["_file"]
        note = ("fp64 transcendental-bound path (SURVEY 8d): the HBM fraction is reported as mandated; valu_frac = "
                "VALU instructions per column-step (rocprofv3 --pmc profile of this code object) x 4 cycles / "
                "(1024 SIMDs x 2.4 GHz x kernel time) is the side that binds; null = no profile of this build")
        if prof is not None and "SQ_INSTS_VALU" in prof and "SQ_WAVES" in prof:
            valu_per_col = prof["SQ_INSTS_VALU"] / prof["SQ_WAVES"]       # one wave per column
            out["valu_instr_per_column_step"] = valu_per_col
            if "SQ_INSTS_SALU" in prof:
                out["salu_instr_per_column_step"] = prof["SQ_INSTS_SALU"] / prof["SQ_WAVES"]
            out["valu_frac"] = valu_per_col * ncol * VALU_CYCLES_PER_INSTR / (N_SIMD * CLOCK_HZ * kern_s)
            if "SQ_ACTIVE_INST_VALU" in prof:
                # the same with the measured issue time of the instructions (SQ_ACTIVE_INST_VALU counts 4-cycle slots;
                # quarter-rate fp64 ops such as v_rcp_f64 take more than one), i.e. the fraction of time the VALUs are busy
                busy = prof["SQ_ACTIVE_INST_VALU"] / prof["SQ_WAVES"] * 4.0
                out["valu_busy_frac"] = busy * ncol / (N_SIMD * CLOCK_HZ * kern_s)
                if "SQ_THREAD_CYCLES_VALU" in prof:
                    # share of the 64 lanes that are active while a VALU instruction issues (EXEC mask): the rest of
                    # the busy time is spent on lanes whose branch is not taken
                    out["valu_lane_util"] = prof["SQ_THREAD_CYCLES_VALU"] / (prof["SQ_ACTIVE_INST_VALU"] * 64.0)
                    out["valu_useful_frac"] = out["valu_busy_frac"] * out["valu_lane_util"]
            # Reachability of the 40 % HBM target (north_star): at PERFECT issue of the present VALU instruction count
            # the kernel would take valu_floor_ms; 40 % of 8 TB/s needs target_ms.
            floor_s = valu_per_col * ncol * VALU_CYCLES_PER_INSTR / (N_SIMD * CLOCK_HZ)
            target_s = algo_bytes * ncol / (0.40 * HBM_PEAK)
            out["valu_floor_ms"] = floor_s * 1e3
            out["frac_at_valu_floor"] = algo_bytes * ncol / floor_s / HBM_PEAK
            out["ms_for_40pct_hbm"] = target_s * 1e3
            note += ("; perfect-issue VALU floor of this instruction count = %.3f ms = %.1f %% of the HBM roofline, the "
                     "40 %% target needs %.3f ms, i.e. %.1fx fewer VALU instructions than today even with no stall at all"
                     % (floor_s * 1e3, 100 * out["frac_at_valu_floor"], target_s * 1e3, floor_s / target_s))
        out["note"] = note
        return out

    ncol = args.ncol or DEFAULT_NCOL[args.workload]
    shard, st, res = time_workload(args.workload, ncol, args.steps, args.warmup)
    out = None
    if rank == 0:
        out = {
            "metric": "thompson_mp_column_steps_per_sec", "value": res["value"], "unit": "column-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"p64": "f64", "p32n": "f32 state / f64 rates", "f32": "f32"}[args.arith],
            "data": "none (launcher rehearsal without a GPU; no physics ran)" if rehearse else "synthetic",
            "config": {"workload": res["workload"], "ncol_per_gpu": ncol, "nz": NZ, "dt": DT,
                       "parallelism": "columns sharded over ranks, no halo, no data-path collective; one RCCL "
                                      "all-gather of the per-rank exact (int64 fixed-point) precipitation accumulators -- the domain "
                                      "means of W:248-303 --, summed locally, inside the timed region: identical sums for every "
                                      "rank count; the optional max-q / negative-value scan runs after the clock"},
            "precip_domain_sums": res["precip_domain_sums"],
            "sanity_max_qc_qr_nr_qs_qi_qg_ni": res["sanity_max_qc_qr_nr_qs_qi_qg_ni"],
            "negative_values": res["negative_values"],
        }
        if not rehearse:
            out["roofline"] = roofline(shard, res)
            out["init_seconds"] = shard.model.init_seconds
    if args.arith != "p64":
        args.no_cpu_baseline = args.no_other_workloads = True   # the accuracy leg and the companion workloads are the p64 build's
    if not rehearse and not args.no_cpu_baseline and world == 1:   # N = 1 only: the host cores are shared by the ranks
        out["accuracy"] = accuracy_leg(shard.model, st, res["iiwarm"], ACCURACY_COLUMNS[args.workload])
        out["cpu_baseline"] = cpu_baseline(st, res["iiwarm"])
        out["gpu_over_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if not rehearse and not args.no_host_entry and not args.no_cpu_baseline and world == 1 and args.arith == "p64":
        out["host_entry"] = host_entry_leg(shard.model, st, res["iiwarm"])
    if not rehearse:
        shard.close()
    del shard, st

    # ---- the other configs of BASELINE.json, same process, after the headline (default run only) ----
    if not rehearse and not args.no_other_workloads and args.workload == "config3" and not args.ncol:
        # config 4 (10^6 columns over 8 GPUs) is represented by ONE of its 125 000-column shards at N = 1 and by a shard per
        # rank at N > 1 (10^6 at N = 8)
        others = ["config2", "config5", "config4"] if world == 1 else ["config4"]
        lines = []
        for name in others:
            shard, st, r = time_workload(name, DEFAULT_NCOL[name], args.steps, min(args.warmup, 3))
            if rank == 0:
                line = {"workload": r["workload"], "value": r["value"], "unit": "column-steps/s",
                        "ms_per_step": r["ms_per_step"], "n_gpus": world, "roofline": roofline(shard, r),
                        "precip_domain_sums": r["precip_domain_sums"], "negative_values": r["negative_values"]}
                if world == 1 and not args.no_cpu_baseline:
                    line["accuracy"] = accuracy_leg(shard.model, st, r["iiwarm"], ACCURACY_COLUMNS[name])
                    if name != "config4":                     # config 4 is config 3's recipe: its CPU baseline is the headline's
                        line["cpu_baseline"] = cpu_baseline(st, r["iiwarm"], budget_s=5.0)
                lines.append(line)
            shard.close()
            del shard, st
        if rank == 0:
            out["other_workloads"] = lines
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # this process stays a pure launcher: no torch, no HIP
    run_rank(args)


if __name__ == "__main__":
    main()
