"""The parity metric of BASELINE.md section 3 / SURVEY.md 8d, plus a conditioning mask.

Why a mask: the reference takes a few decisions on cancellation residues.  The
clearest is M:3595-3596: after block M has evaporated ALL cloud water
(prw_vcd = -rc*orho*odt, M:2854), `xrc = MAX(0., qc1d + qcten*DT)` is
+/- 1e-20 rounding noise, and `if (temp < HGFR .and. xrc > 0.)` then moves the
whole droplet number into cloud ice (M:3598-3602).  One ulp of difference in any
upstream libm call flips that branch and changes n_i by orders of magnitude.
Such levels are chaotic in the reference itself, so no implementation can match
another's choice there -- but each of the TWO outcomes is well defined, and block Q
and block R are pointwise in k.  The oracle therefore flags the levels whose `> 0.`
test sits on a cancellation residue and can be told to take the test as true or as
false there (th_oracle_mp_thompson_force); a flagged level of the HIP result must
equal ONE of the two outcomes, in all variables at once (branch_aware_compare).
No level is left unchecked: a level whose ORACLE output itself moves by more than
the tolerance under ulp-sized input perturbations is held to a multiple of that
measured sensitivity instead of being dropped.
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


def level_err(got, ref, keys=OUT, inp=None, depletion=0.0):
    """[ncol, nz]: max over the variables of |x-ref|/max(|ref|, floor) at each level.  With `inp` and
    `depletion` d the floor of a variable is raised to d*|input|: a species consumed to a fraction < d of its
    input in one call carries the input's rounding error amplified by 1/d (long time steps, wild fuzz inputs)."""
    e = None
    for k in keys:
        fl = FLOORS[k] if not depletion else np.maximum(FLOORS[k], depletion * np.abs(np.asarray(inp[k])))
        r = rel_err(np.asarray(got[k]), np.asarray(ref[k]), fl)
        e = r if e is None else np.maximum(e, r)
    return e


_PERT = [(1 + 2 * _EPS, 1 - 2 * _EPS), (1 - 2 * _EPS, 1 + 2 * _EPS), (1 + 4 * _EPS, 1 + 2 * _EPS),
         (1 - 4 * _EPS, 1 - 2 * _EPS)]


def _copy(st):
    return {k: np.ascontiguousarray(v.copy()) for k, v in st.items()}


def branch_aware_compare(oracle, st, dt, got, got_ppt=None, nperturb=2, keys=OUT, depletion=0.0):
    """Level-by-level comparison of `got` (the HIP result after one step from `st`) with the oracle that leaves
    no level unchecked.  Returns a dict of [ncol, nz] arrays and the oracle outputs:
      err    error against the oracle; at levels on one of the reference's residue-decided tests (M:3587 /
             M:3596) the smaller of the errors against the two admissible outcomes (test forced true / false),
             each taken over ALL variables of the level at once
      flags  the oracle's flags (bit 0: M:3587, bit 1: M:3596)
      sens   the oracle's own response at the level (same metric) to ulp-sized perturbations of T and q
      ref, ppt_ref, ppt_err"""
    ref = _copy(st)
    ppt_ref, flags = oracle.batch_step(ref, dt, want_illcond=True)
    err = level_err(got, ref, keys, st, depletion)
    flagged = flags != 0
    refs = [ref]
    if flagged.any():
        e2 = []
        for force in (1, 2):
            r = _copy(st)
            oracle.batch_step(r, dt, force=force)
            refs.append(r)
            e2.append(level_err(got, r, keys, st, depletion))
        err = np.where(flagged, np.minimum(e2[0], e2[1]), err)
    sens = np.zeros_like(err)
    for ft, fq in _PERT[:nperturb]:
        pert = _copy(st)
        pert["t"] *= ft
        for k in ("qv", "qc", "qi", "qr", "qs", "qg"):
            pert[k] *= fq
        base = _copy(pert)
        _, pflags = oracle.batch_step(pert, dt, want_illcond=True)
        s = level_err(pert, ref, keys, st, depletion)
        both = flagged | (pflags != 0)
        if both.any():                       # residue-decided levels: sensitivity within one outcome, not across the flip
            pt = _copy(base)
            oracle.batch_step(pt, dt, force=1)
            s = np.where(both, level_err(pt, refs[1] if len(refs) > 1 else ref, keys, st, depletion), s)
            if len(refs) == 1:               # flagged only in the perturbed run: compare like with like
                rt = _copy(st)
                oracle.batch_step(rt, dt, force=1)
                s = np.where(both, level_err(pt, rt, keys, st, depletion), s)
        sens = np.maximum(sens, s)
    ppt_err = None
    if got_ppt is not None:
        ppt_err = np.abs(np.asarray(got_ppt) - ppt_ref) / np.maximum(np.abs(ppt_ref), FLOORS["ppt"])
    return dict(err=err, flags=flags, sens=sens, ref=ref, ppt_ref=ppt_ref, ppt_err=ppt_err)


def verdict(cmp, tol=TOL, sens_factor=10.0, sens_cut=1e-11):
    """Summary of a branch_aware_compare: every level must be within max(tol, sens_factor * sens).
      max_rel            worst error over the levels the oracle itself holds steady (sens <= sens_cut), flagged
                         levels included through their matching outcome
      n_unmatched        levels (any kind) beyond max(tol, sens_factor * sens): these are failures
      n_branch_levels    levels on a residue-decided test
      n_sensitive        levels whose oracle output moves by more than sens_cut under ulp perturbations, and
      worst_sensitive    the worst error / sensitivity ratio among them
      max_err_any        worst error over ALL levels (nothing filtered), n_beyond_tol the number of levels beyond
                         `tol`, cols_within_tol_frac the fraction of columns with every level within `tol`"""
    err, sens, flags = cmp["err"], cmp["sens"], cmp["flags"]
    steady = sens <= sens_cut
    lim = np.maximum(tol, sens_factor * sens)
    ratio = np.where(~steady, err / np.maximum(sens, 1e-300), 0.0)
    out = dict(max_rel=float(err[steady].max()) if steady.any() else 0.0,
               n_unmatched=int((err > lim).sum()), n_levels=int(err.size),
               n_branch_levels=int((flags != 0).sum()), n_sensitive=int((~steady).sum()),
               worst_sensitive=float(ratio.max()) if ratio.size else 0.0,
               max_err_any=float(err.max()) if err.size else 0.0,
               # the plain statistics against `tol`, no level left out and no allowance for sensitivity
               n_beyond_tol=int((err > tol).sum()),
               cols_within_tol_frac=float((err <= tol).all(axis=1).mean()) if err.ndim == 2 and err.size else 1.0,
               max_sens=float(sens.max()) if sens.size else 0.0)
    if cmp.get("ppt_err") is not None:
        out["max_rel_ppt"] = float(cmp["ppt_err"].max())
        out["max_rel"] = max(out["max_rel"], out["max_rel_ppt"])
    return out


def branch_aware_max_rel(oracle, st, dt, got, got_ppt=None, tol=TOL):
    """bench.py's accuracy figure: verdict(branch_aware_compare(...))."""
    return verdict(branch_aware_compare(oracle, st, dt, got, got_ppt), tol=tol)


ABS_CEILING = 1e-6          # no level may be further off than this, however sensitive the oracle is there
MAX_SENSITIVE_FRAC = 0.02   # at most this share of the levels may lean on the sensitivity allowance


def assert_parity(oracle, st, dt, got, got_ppt, tol=TOL, sens_factor=10.0, tol_ppt=None, depletion=0.0,
                  max_branch_frac=None, min_cols_within=None, tol_cols=TOL, ceiling=ABS_CEILING,
                  max_sensitive_frac=MAX_SENSITIVE_FRAC, quantile=None):
    """The parity assertion of the -m gpu tests: EVERY level within max(tol, sens_factor x the oracle's own
    sensitivity there), levels on the reference's residue-decided tests against the better of their two admissible
    outcomes; precipitation within tol_ppt.  The sensitivity allowance is itself bounded: no level beyond
    max(tol, ceiling) whatever the oracle's sensitivity there, and at most max_sensitive_frac of the levels beyond
    `tol` at all (i.e. passing only because of the allowance).  Optionally: at most max_branch_frac of the levels on
    residue-decided tests, and at least min_cols_within of the columns with every level within tol_cols; `quantile` =
    (q, bound): the q-quantile of the level errors (no allowance of any kind) must not exceed bound.
    Returns the verdict dict."""
    cmp = branch_aware_compare(oracle, st, dt, got, got_ppt, depletion=depletion)
    v = verdict(cmp, tol=tol, sens_factor=sens_factor)
    assert v["n_unmatched"] == 0, v
    assert v["max_err_any"] <= max(tol, ceiling), v
    assert v["n_beyond_tol"] <= int(np.ceil(max_sensitive_frac * v["n_levels"])), v
    if got_ppt is not None:
        assert v["max_rel_ppt"] < (tol if tol_ppt is None else tol_ppt), v
    if quantile is not None:
        v["quantile"] = float(np.quantile(cmp["err"], quantile[0]))
        assert v["quantile"] <= quantile[1], v
    if max_branch_frac is not None:
        assert v["n_branch_levels"] <= max_branch_frac * v["n_levels"], v
    if min_cols_within is not None:
        lim = np.maximum(tol_cols, sens_factor * cmp["sens"])
        frac = float((cmp["err"] <= lim).all(axis=1).mean())
        v["cols_within"] = frac
        assert frac >= min_cols_within, v
    return v
