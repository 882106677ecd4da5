"""The parity metric of BASELINE.md section 3 / SURVEY.md 8d."""
import numpy as np

TOL = 1e-10
FLOORS = dict(qv=1e-12, qc=1e-12, qi=1e-12, qr=1e-12, qs=1e-12, qg=1e-12,
              ni=1e-6, nr=1e-6, nc=1e-6, nwfa=1e-6, nifa=1e-6, t=1.0, ppt=1e-12)


def rel_err(x, ref, floor):
    return np.abs(x - ref) / np.maximum(np.abs(ref), floor)


def max_rel(got, ref, keys):
    """max over variables/levels/columns of |x-ref|/max(|ref|,floor); returns (max, per-variable dict)."""
    per = {}
    for k in keys:
        per[k] = float(np.max(rel_err(np.asarray(got[k]), np.asarray(ref[k]), FLOORS[k]))) if np.size(ref[k]) else 0.0
    return max(per.values()), per
