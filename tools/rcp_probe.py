import sys; sys.path[:0]=['/root/repo']
import numpy as np, ctypes as C
from kid_amd import ThompsonMP
from kid_amd.thompson import load_library, _np_ptr
m=ThompsonMP(iiwarm=True); L=load_library()
rng=np.random.default_rng(1)
n=2_000_000
b=np.exp(rng.uniform(-40,40,n)); a=np.exp(rng.uniform(-20,20,n))
def probe(fn):
    out=np.empty(n); rc=L.kidmp_math_probe(m._h, fn, n, _np_ptr(a), _np_ptr(b), _np_ptr(out)); assert rc==0; return out
ld=np.longdouble
r=probe(7); e=np.abs((r.astype(ld)*b.astype(ld)-1)); print("rcp raw: max rel err 2^%.2f, mean 2^%.2f"%(np.log2(float(e.max())), np.log2(float(e.mean()))))
ref=(a.astype(ld)/b.astype(ld))
for fn,name in ((8,"fm::div"),(9,"IEEE")):
    q=probe(fn); ulp=np.abs((q.astype(ld)-ref)/ref)/2**-53
    print(name,"max err %.3f ulp(rel 2^-53 units), frac !=IEEE"% float(ulp.max()), float((q!=probe(9)).mean()))
q=probe(10); ulp=np.abs((q.astype(ld)*b.astype(ld)-1))/2**-53
print("fm::rcp: max |b*y-1| = %.3f ulp" % float(ulp.max()))
