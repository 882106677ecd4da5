#!/usr/bin/env python3
"""End-to-end rate of the Fortran drop-in (mini KiD driver -> mphys_thompson09_interfacen -> libkidmp.so).

The driver itself times its step loop (`time=1`: system_clock around `nsteps` calls of mphys_thompson09_interfacen, after
one untimed call that builds the tables and touches the staging memory; the model-side state update is skipped, it is
KiD's cost, not the drop-in's).  Every configuration is run `REPS` times; the median is reported with the spread, so a
figure can never be a difference of two noisy runs (round 2's tool was, and printed a negative time).

usage: fortran_dropin_rate.py [threads ...]     (OpenMP threads of the adapter's gather / back-out loops; default 1 and
                                                 the cores this process may use)"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
exe = os.path.join(ROOT, "kid_amd", "fortran", "build", "kid_mini_driver")
REPS = 3


def run(nx, nsteps, case, rates, threads, extra=()):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
    out = subprocess.run([exe, str(nx), str(nsteps), case, "0", "p64", "-" if rates else "norates", "time=1"] + list(extra),
                         check=True, capture_output=True, text=True, cwd=os.path.dirname(exe), env=env).stdout
    for line in out.splitlines():
        p = line.split()
        if p and p[0] == "TIME":
            sec = float(p[2])
            assert sec > 0, line
            return sec / int(p[1])
    raise RuntimeError("no TIME line:\n" + out)


def main():
    from bench import host_cores
    avail, how = host_cores()
    threads = [int(a) for a in sys.argv[1:]] or sorted({1, avail})
    print("# cores available to this process: %d (%s)" % (avail, how))
    for rates in (False, True):
        for nx, case, nsteps in ((1000, "warm", 40), (10000, "warm", 20), (10000, "mixed", 20), (50000, "mixed", 10)):
            if rates and nx > 10000:
                continue
            for th in threads:
                ts = sorted(run(nx, nsteps, case, rates, th) for _ in range(REPS))
                med = statistics.median(ts)
                print("nx=%6d %-5s rate diagnostics %-3s omp threads %3d: %8.3f ms per step (min %.3f, max %.3f, %d runs of %d steps), "
                      "%.3e column-steps/s" % (nx, case, "on" if rates else "off", th, med * 1e3, ts[0] * 1e3, ts[-1] * 1e3,
                                               REPS, nsteps, nx / med), flush=True)


if __name__ == "__main__":
    main()
