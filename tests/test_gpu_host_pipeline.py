"""The host-array entries (kidmp_batch_step_host*, kidmp32_batch_step_host) as a chunked upload / step / download
pipeline (-m gpu): whatever the chunking and whatever kind of host memory, they must return bit for bit what the
device-resident entry returns for the same columns -- columns are independent, so cutting the batch changes nothing."""
import ctypes as C

import numpy as np
import pytest
import torch

import cases
from kid_amd import thompson
from kid_amd.thompson import NRATES, STATE_NAMES, FORCING_NAMES

pytestmark = pytest.mark.gpu
KEYS = STATE_NAMES + FORCING_NAMES


def _device_reference(m, st, dt, want_rates):
    dev = torch.device("cuda", 0)
    d = {k: torch.as_tensor(st[k]).to(dev) for k in KEYS}
    ncol, nz = st["qv"].shape
    ppt = torch.zeros(ncol, 4, dtype=torch.float64, device=dev)
    rates = torch.zeros(ncol, NRATES, nz, dtype=torch.float64, device=dev) if want_rates else None
    nstep = torch.zeros(ncol, 4, dtype=torch.int32, device=dev)
    m.batch_step(d, dt, ppt, rates=rates, nstep=nstep)
    torch.cuda.synchronize()
    return ({k: d[k].cpu().numpy() for k in STATE_NAMES}, ppt.cpu().numpy(),
            rates.cpu().numpy() if want_rates else None, nstep.cpu().numpy())


def _state(name, ncol):
    st = cases.config3(ncol, seed=cases.SEED + 7) if name == "mixed" else cases.config2(ncol)
    if name == "warm":                                       # replicated columns: make them differ
        rng = np.random.default_rng(5)
        st["qr"] = st["qr"] * rng.uniform(0.5, 2.0, size=(ncol, 1))
    return {k: np.ascontiguousarray(st[k]) for k in KEYS}


@pytest.mark.parametrize("name,chunk,kind", [("mixed", 0, "pageable"), ("mixed", 0, "pinned"), ("mixed", 1024, "pageable"),
                                             ("mixed", 1024, "pinned"), ("mixed", 700, "pageable"), ("mixed", 700, "pinned"),
                                             ("warm", 256, "pageable"), ("warm", 256, "pinned")])
def test_host_entry_equals_device_entry(gpu_mixed, gpu_warm, name, chunk, kind):
    m = gpu_mixed if name == "mixed" else gpu_warm
    ncol = 4500                                              # default chunking: 4 chunks of 1280, the last one ragged
    st = _state(name, ncol)
    ref, ref_ppt, _, _ = _device_reference(m, st, 10.0, False)
    if kind == "pinned":
        got = {k: thompson.host_pinned_copy(st[k]) for k in KEYS}
        ppt = thompson.host_empty((ncol, 4)); ppt[...] = 0.0
    else:
        got = {k: st[k].copy() for k in KEYS}
        ppt = np.zeros((ncol, 4))
    try:
        m.set_host_chunk(chunk)
        m.batch_step_host(got, 10.0, ppt=ppt)
    finally:
        m.set_host_chunk(0)
    for k in STATE_NAMES:
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(ppt, ref_ppt)
    for k in FORCING_NAMES:
        assert np.array_equal(got[k], st[k]), k              # inputs untouched


def test_host_entry_rates_and_substeps_through_the_pipeline(gpu_mixed):
    ncol = 3000
    st = _state("mixed", ncol)
    ref, ref_ppt, ref_rates, ref_nstep = _device_reference(gpu_mixed, st, 10.0, True)
    L = thompson.load_library()
    got = {k: thompson.host_pinned_copy(st[k]) for k in KEYS}
    ppt = thompson.host_empty((ncol, 4)); ppt[...] = 0.0
    rates = thompson.host_empty((ncol, NRATES, 120)); rates[...] = -1.0
    nstep = thompson.host_empty((ncol, 4), np.int32); nstep[...] = -1
    dp = C.POINTER(C.c_double)
    gpu_mixed.set_host_chunk(512)                            # 6 chunks, the last one of 440 columns; the ring wraps twice
    try:
        rc = L.kidmp_batch_step_host_diag(gpu_mixed._h, C.c_int64(ncol), C.c_int32(120), C.c_double(10.0),
                                          *[got[k].ctypes.data_as(dp) for k in KEYS],
                                          ppt.ctypes.data_as(dp), rates.ctypes.data_as(dp),
                                          nstep.ctypes.data_as(C.POINTER(C.c_int32)))
    finally:
        gpu_mixed.set_host_chunk(0)
    assert rc == 0
    for k in STATE_NAMES:
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(ppt, ref_ppt)
    assert np.array_equal(rates, ref_rates)
    assert np.array_equal(nstep, ref_nstep)


def test_host_entry_accumulates_ppt_and_repeats(gpu_warm):
    """ppt is INOUT (M:1172): two calls through the pipeline accumulate like two calls of the device entry."""
    ncol = 2500
    st = _state("warm", ncol)
    dev = torch.device("cuda", 0)
    d = {k: torch.as_tensor(st[k]).to(dev) for k in KEYS}
    dppt = torch.zeros(ncol, 4, dtype=torch.float64, device=dev)
    got = {k: thompson.host_pinned_copy(st[k]) for k in KEYS}
    ppt = thompson.host_empty((ncol, 4)); ppt[...] = 0.0
    for _ in range(3):
        gpu_warm.batch_step(d, 10.0, dppt)
        gpu_warm.batch_step_host(got, 10.0, ppt=ppt)
    torch.cuda.synchronize()
    for k in STATE_NAMES:
        assert np.array_equal(got[k], d[k].cpu().numpy()), k
    assert np.array_equal(ppt, dppt.cpu().numpy())
    assert ppt[:, 0].sum() > 0.0


def test_binary32_host_entry_through_the_pipeline(gpu_mixed):
    ncol = 2600
    st = {k: v.astype(np.float32) for k, v in _state("mixed", ncol).items()}
    dev = torch.device("cuda", 0)
    d = {k: torch.as_tensor(st[k]).to(dev) for k in KEYS}
    dppt = torch.zeros(ncol, 4, dtype=torch.float32, device=dev)
    gpu_mixed.batch_step32(d, 10.0, dppt, arith="p32n")
    torch.cuda.synchronize()
    got = {k: thompson.host_pinned_copy(st[k]) for k in KEYS}
    gpu_mixed.set_host_chunk(1000)
    try:
        ppt, _, nstep = gpu_mixed.batch_step32_host(got, 10.0, arith="p32n", want_nstep=True)
    finally:
        gpu_mixed.set_host_chunk(0)
    for k in STATE_NAMES:
        assert np.array_equal(got[k], d[k].cpu().numpy()), k
    assert np.array_equal(ppt, dppt.cpu().numpy())
    assert (nstep >= 0).all() and nstep.max() >= 1


def test_pinned_block_outlives_its_first_view():
    a = thompson.host_empty((8, 16))
    a[...] = 3.0
    row = a[2]
    del a
    import gc
    gc.collect()
    assert row.sum() == 48.0                                 # the block is still alive: the view holds it


def test_host_chunk_argument_checks(gpu_mixed):
    L = thompson.load_library()
    assert L.kidmp_set_host_chunk(gpu_mixed._h, -1) != 0
    assert L.kidmp_set_host_chunk(None, 16) != 0


def test_aerosol_aware_context_uploads_the_updraft(gpu_mixed_aero):
    """is_aerosol_aware contexts read w (activ_ncloud, M:2797): the pipeline carries it as a 15th profile, chunk by chunk."""
    ncol = 3000
    st = cases.config3(ncol, seed=cases.SEED + 9)
    rng = np.random.default_rng(11)
    st["w"] = rng.uniform(-0.5, 8.0, size=(ncol, 1)) * np.ones_like(st["qv"])
    st["nwfa"] = st["nwfa"] * rng.uniform(0.3, 30.0, size=(ncol, 1))
    st = {k: np.ascontiguousarray(st[k]) for k in KEYS}
    ref, ref_ppt, _, _ = _device_reference(gpu_mixed_aero, st, 10.0, False)
    got = {k: thompson.host_pinned_copy(st[k]) for k in KEYS}
    gpu_mixed_aero.set_host_chunk(640)                       # five chunks, the last one ragged
    try:
        ppt, _ = gpu_mixed_aero.batch_step_host(got, 10.0)
    finally:
        gpu_mixed_aero.set_host_chunk(0)
    for k in STATE_NAMES:
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(ppt, ref_ppt)
    # and w matters: another updraft, another droplet number
    other = {k: st[k].copy() for k in KEYS}
    other["w"] = other["w"] + 2.0
    gpu_mixed_aero.batch_step_host(other, 10.0)
    assert not np.array_equal(other["nc"], got["nc"])
    # a missing updraft is refused, not read from address 0
    L = thompson.load_library()
    dp = C.POINTER(C.c_double)
    args = [got[k].ctypes.data_as(dp) for k in KEYS]
    args[KEYS.index("w")] = None
    rc = L.kidmp_batch_step_host(gpu_mixed_aero._h, C.c_int64(ncol), C.c_int32(120), C.c_double(10.0), *args,
                                 ppt.ctypes.data_as(dp), None)
    assert rc != 0 and b"updraft" in L.kidmp_last_error(gpu_mixed_aero._h)


def test_arrays_kid_never_fills_may_be_left_out(gpu_warm, gpu_mixed):
    """nc, nwfa, nifa (W:36 passes them unset) and, in a warm run, the frozen species: left out of the host-array call they
    neither cross PCIe nor come back, and the step is the one the full call makes with the defaults / zeros in place."""
    ncol = 3000
    st = _state("warm", ncol)
    for k in ("qi", "qs", "qg", "ni"):
        st[k][...] = 0.0
    d = {k: torch.as_tensor(st[k]).cuda() for k in KEYS}
    nc, nwfa, nifa = gpu_warm.default_aerosols(d["qv"], d["t"], d["p"])
    full = {k: st[k].copy() for k in KEYS}
    full["nc"], full["nwfa"], full["nifa"] = nc.cpu().numpy(), nwfa.cpu().numpy(), nifa.cpu().numpy()
    ppt_full, _ = gpu_warm.batch_step_host(full, 10.0)
    lean = {k: thompson.host_pinned_copy(st[k]) for k in ("qv", "qc", "qr", "nr", "t", "p", "dz")}
    gpu_warm.set_host_chunk(1024)
    try:
        ppt_lean, _ = gpu_warm.batch_step_host(lean, 10.0)
    finally:
        gpu_warm.set_host_chunk(0)
    for k in ("qv", "qc", "qr", "nr", "t"):
        assert np.array_equal(lean[k], full[k]), k
    assert np.array_equal(ppt_lean, ppt_full) and ppt_lean[:, 0].sum() > 0
    # mixed-phase context: only the aerosol arrays may be missing
    stm = _state("mixed", 1500)
    dm = {k: torch.as_tensor(stm[k]).cuda() for k in KEYS}
    nc, nwfa, nifa = gpu_mixed.default_aerosols(dm["qv"], dm["t"], dm["p"])
    fullm = {k: stm[k].copy() for k in KEYS}
    fullm["nc"], fullm["nwfa"], fullm["nifa"] = nc.cpu().numpy(), nwfa.cpu().numpy(), nifa.cpu().numpy()
    pf, _ = gpu_mixed.batch_step_host(fullm, 10.0)
    leanm = {k: stm[k].copy() for k in KEYS if k not in ("nc", "nwfa", "nifa", "w")}
    pl, _ = gpu_mixed.batch_step_host(leanm, 10.0)
    for k in leanm:
        assert np.array_equal(leanm[k], fullm[k]), k
    assert np.array_equal(pl, pf)
    # what a context reads cannot be left out, and the groups go together
    from kid_amd import KidmpError
    with pytest.raises(KidmpError, match="mixed-phase context needs"):
        gpu_mixed.batch_step_host({k: stm[k].copy() for k in ("qv", "qc", "qr", "nr", "t", "p", "dz")}, 10.0)
    with pytest.raises(KidmpError, match="together"):
        gpu_mixed.batch_step_host({k: stm[k].copy() for k in KEYS if k != "nwfa"}, 10.0)


def test_aerosol_aware_context_refuses_missing_aerosols(gpu_mixed_aero):
    from kid_amd import KidmpError
    st = _state("mixed", 64)
    with pytest.raises(KidmpError, match="aerosol-aware context needs"):
        gpu_mixed_aero.batch_step_host({k: st[k].copy() for k in KEYS if k not in ("nc", "nwfa", "nifa")}, 10.0)


@pytest.mark.parametrize("nz,ncol", [(77, 2300), (200, 2100), (120, 1), (33, 5000)])
def test_other_shapes_through_the_pipeline(gpu_mixed, nz, ncol):
    """Other level counts (other kernel instantiations, other slice sizes) and degenerate batches."""
    base = cases.config3(ncol, seed=cases.SEED + 3)
    if nz <= 120:
        st = {k: np.ascontiguousarray(base[k][:, :nz]) for k in KEYS}
    else:                                                    # taller column: the profile repeated upwards
        reps = (nz + 119) // 120
        st = {k: np.ascontiguousarray(np.tile(base[k], (1, reps))[:, :nz]) for k in KEYS}
    ref, ref_ppt, _, _ = _device_reference(gpu_mixed, st, 10.0, False)
    got = {k: thompson.host_pinned_copy(st[k]) for k in KEYS}
    ppt, _ = gpu_mixed.batch_step_host(got, 10.0)
    for k in STATE_NAMES:
        assert np.array_equal(got[k], ref[k]), k
    assert np.array_equal(ppt, ref_ppt)


def test_empty_batch_is_a_no_op(gpu_mixed):
    st = {k: np.zeros((0, 120)) for k in KEYS}
    ppt, _ = gpu_mixed.batch_step_host(st, 10.0)
    assert ppt.shape == (0, 4)
