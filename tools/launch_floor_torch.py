import torch, time
x = torch.zeros(64, device="cuda")
big = torch.zeros(10000*256, device="cuda")
for t, name in ((x, "64-element add_"), (big, "2.56M-element add_ (10000 blocks of 256)")):
    for _ in range(20): t.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): t.add_(1.0)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.2f us per launch" % (e0.elapsed_time(e1) / 200 * 1e3))
