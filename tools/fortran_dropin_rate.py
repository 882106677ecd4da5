#!/usr/bin/env python3
"""End-to-end rate of the Fortran drop-in (mini KiD driver -> mphys_thompson09_interfacen -> libkidmp.so): the time of
20 extra steps (a 40-step run minus a 20-step run, which cancels start-up and table building), nx columns each."""
import os
import subprocess
import time

exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kid_amd", "fortran", "build", "kid_mini_driver")


def run(nx, nsteps, case, rates):
    t0 = time.perf_counter()
    subprocess.run([exe, str(nx), str(nsteps), case, "0", "p64"] + ([] if rates else ["norates"]), check=True,
                   stdout=subprocess.DEVNULL, cwd=os.path.dirname(exe))
    return time.perf_counter() - t0


for rates in (True, False):
    for nx, case in ((1000, "warm"), (10000, "warm"), (10000, "mixed"), (50000, "mixed")):
        d = (run(nx, 40, case, rates) - run(nx, 20, case, rates)) / 20
        print("nx=%6d %-5s rate diagnostics %-3s: %8.3f ms per step, %.3e column-steps/s"
              % (nx, case, "on" if rates else "off", d * 1e3, nx / d), flush=True)
