import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, cases, kat_cases as kc
from kid_amd import ThompsonMP
from oracle.oracle import Oracle
from parity import FLOORS, OUT
f32=np.float32
m=ThompsonMP(iiwarm=True); o=Oracle(iiwarm=True)
st={k:np.ascontiguousarray(v.astype(f32)) for k,v in cases.config2(64).items()}
st["qr"]*=np.linspace(0.5,1.5,64,dtype=f32)[:,None]
ref={k:v.copy() for k,v in st.items()}; o.batch_step_p32n(ref,10.0)
got={k:v.copy() for k,v in st.items()}; m.batch_step32_host(got,10.0,arith="p32n")
for k in OUT:
    a,b=got[k].astype(np.float64),ref[k].astype(np.float64)
    e=np.abs(a-b)/np.maximum(np.abs(b),1e4*FLOORS[k])
    sgn=np.mean((a-b)/np.maximum(np.abs(b),1e4*FLOORS[k]))
    print(k,"median %.1e q99 %.1e max %.1e mean signed %.2e"%(np.median(e),np.quantile(e,0.99),e.max(),sgn))
# multi-step drift of the sums, column 0 of KAT-A warm
for nsteps in (30,90):
    s0={k:np.ascontiguousarray(v.astype(f32)[None,:]) for k,v in cases.warm_column_t0().items()}
    a={k:v.copy() for k,v in s0.items()}; b={k:v.copy() for k,v in s0.items()}
    for n in range(nsteps):
        m.batch_step32_host(a,10.0,arith="p32n"); o.batch_step_p32n(b,10.0)
    print(nsteps,"steps: sum qc gpu %.7e oracle %.7e | qr %.7e %.7e | nr %.7e %.7e"%(a["qc"].sum(dtype=np.float64),b["qc"].sum(dtype=np.float64),a["qr"].sum(dtype=np.float64),b["qr"].sum(dtype=np.float64),a["nr"].sum(dtype=np.float64),b["nr"].sum(dtype=np.float64)))
