"""Caller-registered host memory under the host-array entry (-m gpu), following the rule of INTEGRATION.md section 4:
ONE page-aligned region that holds every array of the call, registered once (hipHostRegister) before the call and
unregistered once after it has returned.

Background: round 2 had kidmp_host_register / kidmp_host_unregister and a test that registered fifteen separately
malloc'ed numpy buffers one by one; a run of that sequence ended in a runtime abort once, no log was kept, and the API
was removed.  That test broke the rule in three ways (non-page-aligned starts; buffers that glibc places back to back
in the brk heap once its dynamic mmap threshold has risen, so that neighbouring registrations share boundary pages;
unregistration under `assert` inside `finally`, which leaves ranges registered when numpy frees them).  This file keeps
the supported pattern under test; it lives in its own file so that, should the runtime abort the process, the log names
exactly this sequence."""
import mmap

import numpy as np
import pytest
import torch

import cases
from kid_amd.thompson import FORCING_NAMES, STATE_NAMES

pytestmark = pytest.mark.gpu
KEYS = STATE_NAMES + FORCING_NAMES


def test_one_page_aligned_registered_region(gpu_mixed):
    ncol, nz = 4500, 120
    st = cases.config3(ncol, seed=cases.SEED + 7)
    dev = torch.device("cuda", 0)
    d = {k: torch.as_tensor(np.ascontiguousarray(st[k])).to(dev) for k in KEYS}
    dppt = torch.zeros(ncol, 4, dtype=torch.float64, device=dev)
    gpu_mixed.batch_step(d, 10.0, dppt)
    torch.cuda.synchronize()

    per = ncol * nz * 8
    per_al = (per + 4095) // 4096 * 4096                     # every array starts on its own page
    total = per_al * len(KEYS) + 4096 * ((ncol * 32 + 4095) // 4096)
    region = mmap.mmap(-1, total)                            # anonymous, page-aligned, a multiple of the page size
    whole = np.frombuffer(region, dtype=np.uint8)
    got = {}
    for n, k in enumerate(KEYS):
        got[k] = whole[n * per_al:n * per_al + per].view(np.float64).reshape(ncol, nz)
        got[k][...] = st[k]
    ppt = whole[len(KEYS) * per_al:len(KEYS) * per_al + ncol * 32].view(np.float64).reshape(ncol, 4)
    ppt[...] = 0.0
    base = whole.ctypes.data
    assert base % 4096 == 0
    rt = torch.cuda.cudart()
    assert int(rt.cudaHostRegister(base, total, 0)) == 0     # hipHostRegister: one region, once
    rc_unreg = None
    try:
        gpu_mixed.set_host_chunk(1024)
        gpu_mixed.batch_step_host(got, 10.0, ppt=ppt)        # returns after its last download has completed
    finally:
        gpu_mixed.set_host_chunk(0)
        rc_unreg = int(rt.cudaHostUnregister(base))           # after the call, no assert in between
    assert rc_unreg == 0
    for k in STATE_NAMES:
        assert np.array_equal(got[k], d[k].cpu().numpy()), k
    assert np.array_equal(ppt, dppt.cpu().numpy())
    del got, ppt, whole
    region.close()
