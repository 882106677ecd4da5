#!/usr/bin/env python3
"""Occupancy probe: times experiment builds of the column kernel (tools/build_variant.sh ... -DKIDMP_EXP_NL=<nz>) on the
first <nz> levels of a BASELINE workload.  A shorter column leaves LDS room for a third workgroup per CU, so builds that
differ only in their register budget (-DKIDMP_MIXED_WAVES=2|3) show what a third wave per SIMD buys the mixed-phase kernel.

usage: occupancy_probe.py --nz 78 libA.so libB.so ... [--workload config3] [--ncol N]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from kid_amd.thompson import _Cfg, STATE_NAMES

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--nz", type=int, default=78)
ap.add_argument("--workload", default="config3")
ap.add_argument("--ncol", type=int, default=100000)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
vp = C.c_void_p

def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.kidmp_init.restype = C.c_int; L.kidmp_init.argtypes = [C.POINTER(_Cfg), C.POINTER(vp)]
    L.kidmp_finalize.argtypes = [vp]
    L.kidmp_last_error.restype = C.c_char_p; L.kidmp_last_error.argtypes = [vp]
    L.kidmp_batch_step_device.restype = C.c_int
    L.kidmp_batch_step_device.argtypes = [vp, C.c_int64, C.c_int32, C.c_double] + [vp] * 18 + [vp]
    return L

libs = [load(p) for p in args.libs]
dev = torch.device("cuda", 0)
st, iiwarm, desc = bench.make_workload(args.workload, args.ncol)
st = {k: np.ascontiguousarray(v[:, :args.nz]) for k, v in st.items()}
ctxs = []
for L in libs:
    cfg = _Cfg(int(iiwarm), 1, 100.0, 0, 0); h = vp()
    assert L.kidmp_init(C.byref(cfg), C.byref(h)) == 0, L.kidmp_last_error(None)
    ctxs.append(h)
fresh = lambda: {k: torch.from_numpy(v).to(dev) for k, v in st.items()}
def step(i, d, ppt):
    s = torch.cuda.current_stream().cuda_stream
    a = [d[k].data_ptr() for k in STATE_NAMES] + [d["p"].data_ptr(), None, d["dz"].data_ptr(), ppt.data_ptr(), None, None]
    rc = libs[i].kidmp_batch_step_device(ctxs[i], args.ncol, args.nz, 10.0, *a, s)
    assert rc == 0, libs[i].kidmp_last_error(ctxs[i])
outs = []
for i in range(len(libs)):
    d, ppt = fresh(), torch.zeros(args.ncol, 4, dtype=torch.float64, device=dev)
    step(i, d, ppt); torch.cuda.synchronize()
    outs.append({k: d[k].cpu().numpy() for k in STATE_NAMES})
for i in range(1, len(libs)):
    print("%s vs first: bit-identical after one step: %s" % (os.path.basename(args.libs[i]), all(np.array_equal(outs[i][k], outs[0][k]) for k in STATE_NAMES)))
states = [(fresh(), torch.zeros(args.ncol, 4, dtype=torch.float64, device=dev)) for _ in libs]
for i in range(len(libs)):
    for _ in range(3): step(i, *states[i])
torch.cuda.synchronize()
times = [[] for _ in libs]
for rep in range(args.reps):
    for i in range(len(libs)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps): step(i, *states[i])
        e1.record(); torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) / args.steps)
base = sorted(times[0])[len(times[0]) // 2]
for i, p in enumerate(args.libs):
    t = sorted(times[i])
    print("%s nz=%d %-32s median %.4f ms  min %.4f  max %.4f  (vs first: %+.2f %%)" % (args.workload, args.nz, os.path.basename(p), t[len(t)//2], t[0], t[-1], 100.0 * (t[len(t)//2] / base - 1)))
