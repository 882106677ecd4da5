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

from parity import FLOORS, OUT, level_err, rel_err


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
            out.append(rec)
    return out
