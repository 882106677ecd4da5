#!/usr/bin/env python3
"""How much of KAT-A mixed NATIVE (sum qi 3.76812e-4, rain at call 200 1.708911e-2) can the lookup tables explain?
The native build forms its table builders' constants in REAL (M:452-553 feed M:3698-4439), the oracle's P32n mode reuses the P64
tables: ~1e-7 relative apart.  This perturbs every table of a P32n oracle context by a random +-1e-7 (and, last line, rounds
them to binary32) and repeats the 200-call run from the binary32-formed inputs.  CPU only, ~1 minute.
    python tools/p32n_table_sensitivity.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kat_cases as kc
import oracle.oracle as om

NAMES = ("tcg_racg tmr_racg tcr_gacr tmg_gacr tnr_racg tnr_gacr tcs_racs1 tmr_racs1 tcs_racs2 tmr_racs2 tcr_sacr1 tms_sacr1 tcr_sacr2 "
         "tms_sacr2 tnr_racs1 tnr_racs2 tnr_sacr1 tnr_sacr2 tpi_qcfz tni_qcfz tpi_qrfz tpg_qrfz tni_qrfz tnr_qrfz tps_iaus tni_iaus "
         "tpi_ide t_Efrw t_Efsw").split()
o = om.Oracle(iiwarm=False)


def view(name):                      # the context's own storage (Oracle.table returns a copy)
    nd = C.c_int(); dims = (C.c_int * 4)()
    p = om.lib().th_oracle_table(o._h, name.encode(), C.byref(nd), dims)
    return np.ctypeslib.as_array(p, shape=(int(np.prod([dims[i] for i in range(nd.value)])),))


def run():
    st = kc.kat_a_native(True)
    for _ in range(200):
        ppt, _, _, _ = o.column_step_p32n(st, 10.0)
    return float(st["qi"].astype(np.float64).sum()), float(ppt[0])


print("native (SURVEY 9h)     sum qi 3.768120e-04  rain 1.708911e-02")
print("P64 tables             sum qi %.6e  rain %.6e" % run())
saved = {n: view(n).copy() for n in NAMES}
rng = np.random.default_rng(1)
for t in range(3):
    for n in NAMES:
        v = view(n); v[:] = saved[n] * (1 + rng.uniform(-1e-7, 1e-7, size=v.shape))
    print("tables x (1 +- 1e-7) #%d sum qi %.6e  rain %.6e" % ((t,) + run()))
for n in NAMES:
    view(n)[:] = saved[n].astype(np.float32).astype(np.float64)
print("tables -> binary32     sum qi %.6e  rain %.6e" % run())
o.close()
