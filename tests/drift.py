"""Drift "after the full run" (SURVEY 8d): a device-resident HIP chain against an independent oracle chain over many
coupled steps, with the branch history tracked.

Two chains start from the same columns: h(n+1) = HIP(h(n)) on the GPU (the state never leaves HBM except as a copy for
the checks), o(n+1) = oracle(o(n)) on the CPU.  After every step the oracle is also run ONE step from the HIP state
h(n): a column whose HIP step differs from that by more than `flip_tol` at some level has taken a decision the oracle
does not take from the same input (one of the reference's residue-decided `> 0.` tests M:3587 / M:3596, a table bin or a
size-limit threshold crossed by an ulp) -- from then on the column is "off the oracle's branch history" and its end
state is reported separately.  Reported at every mark (e.g. 60 and 360 steps), per variable: median, 99th percentile
and maximum over the levels of |h - o| / max(|o|, floor), over all columns and over the columns still on the history."""
import numpy as np

from parity import FLOORS, OUT, _EPS, branch_aware_compare, level_err, rel_err

# Why did a column leave the oracle's branch history?  Every departure is classified from the one-step pair
# (h_prev -> h) that caused it:
#   "residue"        every level of the HIP step equals ONE of the two admissible outcomes of the reference's
#                    residue-decided tests (M:3587 / M:3596) -- the other outcome than the oracle's own
#   "sensitivity"    at the levels that differ, the oracle's OWN output moves by at least a tenth of the difference when
#                    its inputs are perturbed by 2-4 ulp (conditioning: near-total depletion, saturation adjustment)
#   "discontinuity"  the same under perturbations of up to LADDER[-1] ulp (2e-13 relative): a table bin, size limit or
#                    threshold of the scheme lies within that distance of the input, and the two implementations,
#                    whose intermediate values agree to a few ulp, stand on different sides of it
#   "unexplained"    none of the above: an arithmetic difference the reference's own conditioning does not account for.
#                    tests/test_gpu_drift.py asserts that there is none.
LADDER = (16, 64, 256, 1024)


def _sub(st, cols):
    return {k: np.ascontiguousarray(v[cols].copy()) for k, v in st.items()}


def explain_departures(oracle, h_prev, h, cols, dt, flip_tol):
    """One record per departing column: {"column", "category", "worst_var", "level", "err", "oracle_response", "ulps"}."""
    cols = np.asarray(cols)
    if cols.size == 0:
        return []
    sp, sh = _sub(h_prev, cols), _sub(h, cols)
    cmp = branch_aware_compare(oracle, sp, dt, sh)
    err, sens, flags, ref = cmp["err"], cmp["sens"], cmp["flags"], cmp["ref"]
    recs = []
    todo = []
    for c in range(cols.size):
        bad = err[c] > flip_tol
        k_bad = int(np.argmax(err[c]))
        per_var = {k: float(rel_err(sh[k][c, k_bad], ref[k][c, k_bad], FLOORS[k])) for k in OUT}
        rec = {"column": int(cols[c]), "level": k_bad, "worst_var": max(per_var, key=per_var.get), "err": float(err[c].max()),
               "on_residue_test": bool((flags[c] != 0).any()), "oracle_response": float(sens[c, k_bad]), "ulps": 4}
        if not bad.any():
            rec["category"] = "residue"
        elif (sens[c][bad] >= 0.1 * err[c][bad]).all():
            rec["category"] = "sensitivity"
        else:
            rec["category"] = None
            todo.append(c)
        recs.append(rec)
    for u in LADDER:                                            # larger perturbations, only for what is still open
        if not todo:
            break
        sub = _sub(sp, todo)
        resp = np.zeros((len(todo), sp["qv"].shape[1]))
        base = _sub(ref, todo)
        for ft, fq in ((1 + u * _EPS, 1 - u * _EPS), (1 - u * _EPS, 1 + u * _EPS), (1 + u * _EPS, 1 + u * _EPS), (1 - u * _EPS, 1 - u * _EPS)):
            pert = {k: v.copy() for k, v in sub.items()}
            pert["t"] *= ft
            for k in ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr"):
                pert[k] *= fq
            oracle.batch_step(pert, dt)
            resp = np.maximum(resp, level_err(pert, base, OUT))
        still = []
        for i, c in enumerate(todo):
            bad = err[c] > flip_tol
            # a discontinuity shows up at the level itself or (through sedimentation) at the levels below it: take the column
            if resp[i].max() >= 0.1 * err[c].max() and (resp[i][bad] >= 0.1 * err[c][bad]).mean() >= 0.5:
                recs[c].update(category="discontinuity", oracle_response=float(resp[i].max()), ulps=u)
            else:
                still.append(c)
        todo = still
    for c in todo:
        recs[c]["category"] = "unexplained"
    return recs


def _copy(st):
    return {k: np.ascontiguousarray(v.copy()) for k, v in st.items()}


def run_chains(model, oracle, st, dt, marks, flip_tol=1e-7):
    import torch
    marks = sorted(marks)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
    ncol = st["qv"].shape[0]
    ppt = torch.zeros(ncol, 4, dtype=torch.float64, device="cuda")
    o = _copy(st)
    oppt = np.zeros((ncol, 4))
    on_hist = np.ones(ncol, dtype=bool)
    left_at = np.zeros(ncol, dtype=np.int64)
    flagged_cols = np.zeros(ncol, dtype=bool)
    departures = []
    out = []
    h = {k: dev[k].cpu().numpy() for k in st}
    for n in range(1, marks[-1] + 1):
        model.batch_step(dev, dt, ppt)
        h_prev, h = h, {k: dev[k].cpu().numpy() for k in st}            # (synchronises)
        # the oracle's own step from the HIP state: same decisions <=> small one-step difference
        chk = _copy(h_prev)
        _, flags = oracle.batch_step(chk, dt, want_illcond=True)
        flagged_cols |= (flags != 0).any(axis=1)
        step_err = level_err(h, chk, OUT).max(axis=1)
        newly = on_hist & (step_err > flip_tol)
        left_at[newly] = n
        on_hist &= ~newly
        for d in explain_departures(oracle, h_prev, h, np.nonzero(newly)[0], dt, flip_tol):
            departures.append(dict(d, step=n))
        oppt += oracle.batch_step(o, dt)
        if n in marks:
            rec = {"steps": n, "columns": int(ncol), "columns_on_oracle_branch_history": int(on_hist.sum()),
                   "frac_on_history": float(on_hist.mean()),
                   "columns_that_met_a_residue_decided_test": int(flagged_cols.sum()), "vars": {}}
            for k in OUT:
                e = rel_err(h[k], o[k], FLOORS[k])
                sel = e[on_hist] if on_hist.any() else e[:0]
                rec["vars"][k] = {"median": float(np.median(e)), "p99": float(np.percentile(e, 99)), "max": float(e.max()),
                                  "on_history": {"median": float(np.median(sel)) if sel.size else 0.0,
                                                 "p99": float(np.percentile(sel, 99)) if sel.size else 0.0,
                                                 "max": float(sel.max()) if sel.size else 0.0}}
            pe = np.abs(ppt.cpu().numpy() - oppt) / np.maximum(np.abs(oppt), FLOORS["ppt"])
            rec["precip_accumulated"] = {"max": float(pe.max()), "max_on_history": float(pe[on_hist].max()) if on_hist.any() else 0.0}
            rec["first_departures_at_steps"] = sorted(int(x) for x in left_at[left_at > 0])[:10]
            cats = [d["category"] for d in departures]
            rec["departures"] = {c: cats.count(c) for c in ("residue", "sensitivity", "discontinuity", "unexplained")}
            rec["departures_not_residue"] = [d for d in departures if d["category"] != "residue"][:20]
            out.append(rec)
    return out
