import sys, os, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); os.chdir('/root/repo')
import bench
from kid_amd import ThompsonMP
st, iiwarm, _ = bench.make_workload("config3", 100000)
m = ThompsonMP(iiwarm=False)
def run(tag, idle=0.0):
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in st.items()}
    ppt = torch.zeros(100000, 4, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    if idle: time.sleep(idle)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for i in range(60):
        m.batch_step(d, 10.0, ppt); ev[i+1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i+1]) for i in range(60)]
    print(tag, "steps 1-5: %s | 6-25 mean %.4f | 26-45 mean %.4f | 46-60 mean %.4f" % (" ".join("%.3f"%x for x in t[:5]), np.mean(t[5:25]), np.mean(t[25:45]), np.mean(t[45:])))
run("first (cold)"); run("again (hot)"); run("after 5 s idle", 5.0); run("again (hot)")
