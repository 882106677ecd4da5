#!/usr/bin/env python3
"""Writes tests/golden/kat_native_inputs.npz: the KAT-B / KAT-A input profiles formed in binary32 arithmetic as a default-REAL
Fortran probe forms them (tests/kat_cases.py: kat_b_native, kat_a_native; powf / expf of this host's libm).  Data only: the
recipes are SURVEY.md 9h's.  Run on the image the native digits were matched on (glibc 2.35)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kat_cases as kc

b, a = kc.kat_b_native(), kc.kat_a_native(True)
np.savez_compressed(os.path.join(HERE, "kat_native_inputs.npz"), b_theta=b["theta"], b_exner=b["exner"], b_qv=b["qv"],
                    b_r_on_cp=np.float32(b["r_on_cp"]), a_t=a["t"], a_p=a["p"], a_qv=a["qv"], a_nc=a["nc"])
print("written")
