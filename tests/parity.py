"""The parity metric of BASELINE.md section 3 / SURVEY.md 8d, plus a conditioning mask.

Why a mask: the reference takes a few decisions on cancellation residues.  The
clearest is M:3595-3596: after block M has evaporated ALL cloud water
(prw_vcd = -rc*orho*odt, M:2854), `xrc = MAX(0., qc1d + qcten*DT)` is
+/- 1e-20 rounding noise, and `if (temp < HGFR .and. xrc > 0.)` then moves the
whole droplet number into cloud ice (M:3598-3602).  One ulp of difference in any
upstream libm call flips that branch and changes n_i by orders of magnitude.
Such levels are chaotic in the reference itself, so no implementation can match
another there; they are detected by re-running the ORACLE on inputs perturbed
by a few ulp and are excluded from the max (and counted).
"""
import numpy as np

TOL = 1e-10
FLOORS = dict(qv=1e-12, qc=1e-12, qi=1e-12, qr=1e-12, qs=1e-12, qg=1e-12,
              ni=1e-6, nr=1e-6, nc=1e-6, nwfa=1e-6, nifa=1e-6, t=1.0, ppt=1e-12)
OUT = ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr", "nc", "nwfa", "nifa", "t")
_EPS = 2.220446049250313e-16


def rel_err(x, ref, floor):
    return np.abs(x - ref) / np.maximum(np.abs(ref), floor)


def max_rel(got, ref, keys=OUT, mask=None):
    """max over variables/levels/columns of |x-ref|/max(|ref|,floor); returns (max, per-variable dict)."""
    per = {}
    for k in keys:
        e = rel_err(np.asarray(got[k]), np.asarray(ref[k]), FLOORS[k])
        if mask is not None:
            e = np.where(mask, e, 0.0)
        per[k] = float(np.max(e)) if e.size else 0.0
    return max(per.values()), per


def conditioned_mask(oracle, st, dt, ref, nperturb=2, thresh=1e-7):
    """True where the reference map is well conditioned at this input.  Two detectors:
    (1) the oracle's own flags for the two `> 0.` tests of M:3587/M:3596 taken on a
        cancellation residue (deterministic);
    (2) generic: ulp-sized input perturbations move no output of the level by more than
        `thresh` (a well-conditioned level moves by ~1e-13)."""
    ncol, nz = st["qv"].shape
    probe = {k: np.ascontiguousarray(v.copy()) for k, v in st.items()}
    _, flags = oracle.batch_step(probe, dt, want_illcond=True)
    ok = flags == 0
    fac = [(1 + 2 * _EPS, 1 - 2 * _EPS), (1 - 2 * _EPS, 1 + 2 * _EPS), (1 + 4 * _EPS, 1 + 2 * _EPS),
           (1 - 4 * _EPS, 1 - 2 * _EPS), (1 + 6 * _EPS, 1 - 4 * _EPS), (1 - 6 * _EPS, 1 + 4 * _EPS)]
    for ft, fq in fac[:nperturb]:
        pert = {k: np.ascontiguousarray(v.copy()) for k, v in st.items()}
        pert["t"] *= ft
        for k in ("qv", "qc", "qi", "qr", "qs", "qg"):
            pert[k] *= fq
        oracle.batch_step(pert, dt)
        for k in OUT:
            ok &= rel_err(pert[k], ref[k], FLOORS[k]) < thresh
    return ok
