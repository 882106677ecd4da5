import sys, os, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.chdir('/root/repo')
import kid_amd
from test_gpu_fuzz import fuzz_columns
from parity import OUT, FLOORS, branch_aware_compare, rel_err
from oracle.oracle import Oracle
lib = sys.argv[1] if len(sys.argv) > 1 else None
kid_amd.load_library(lib)
from kid_amd import ThompsonMP
m = ThompsonMP(iiwarm=False); o = Oracle(iiwarm=False, nthreads=16)
st = fuzz_columns(400, 120, 1)
got = {k: v.copy() for k, v in st.items()}
gppt, _ = m.batch_step_host(got, 10.0)
cmp = branch_aware_compare(o, st, 10.0, got, gppt, depletion=1e-5)
err, sens, flags, ref = cmp["err"], cmp["sens"], cmp["flags"], cmp["ref"]
lim = np.maximum(1e-10, 10*sens)
bad = np.argwhere(err > lim)
print(lib, "levels beyond:", len(bad), "columns:", len(set(bad[:,0])))
for c,k in bad[:40]:
    per = {v: float(rel_err(got[v][c,k], ref[v][c,k], max(FLOORS[v], 1e-5*abs(st[v][c,k])))) for v in OUT}
    w = max(per, key=per.get)
    print(c, k, "worst", w, "%.2e"%per[w], "sens %.1e"%sens[c,k], "flags", flags[c,k], "T %.2f"%st["t"][c,k], "in: qc %.2e qi %.2e qr %.2e qs %.2e qg %.2e | out %s got %.6e ref %.6e" % (st["qc"][c,k], st["qi"][c,k], st["qr"][c,k], st["qs"][c,k], st["qg"][c,k], w, got[w][c,k], ref[w][c,k]))
