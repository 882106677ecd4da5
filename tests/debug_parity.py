"""Debug helper (not a test): prints the worst GPU-vs-oracle mismatches per case."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cases
from parity import FLOORS
from oracle.oracle import Oracle
from kid_amd import ThompsonMP

OUT = ("qv", "qc", "qi", "qr", "qs", "qg", "ni", "nr", "nc", "nwfa", "nifa", "t")


def report(name, st, o, m, dt=10.0, top=6):
    ref = {k: v.copy() for k, v in st.items()}
    rppt = o.batch_step(ref, dt)
    got = {k: v.copy() for k, v in st.items()}
    gppt, _ = m.batch_step_host(got, dt)
    rows = []
    for k in OUT:
        e = np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), FLOORS[k])
        idx = np.argsort(e.ravel())[::-1][:top]
        for i in idx:
            c, lev = np.unravel_index(i, e.shape)
            if e[c, lev] > 1e-11:
                rows.append((e[c, lev], k, c, lev, got[k][c, lev], ref[k][c, lev], st[k][c, lev], st["t"][c, lev]))
    rows.sort(reverse=True)
    print("==", name, "worst:")
    for r in rows[:top * 3]:
        print("  err %.3e %-4s col %d k %d got %.17g ref %.17g in %.17g T %.3f" % r)
    print("  ppt max abs diff", np.abs(gppt - rppt).max())


if __name__ == "__main__":
    o = Oracle(iiwarm=False)
    m = ThompsonMP(iiwarm=False)
    report("edge", cases.edge_cases(), o, m)
    report("config3", cases.config3(512), o, m)
    report("config5", cases.config5(256), o, m)
