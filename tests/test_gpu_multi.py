"""The in-library multi-GPU entry (-m gpu): kidmp_init_multi / kidmp_batch_step_host_multi, what a Fortran or C host
reaches without MPI (the `do i=1,nx` loop of W:54-246 over a device list, the nx-means of W:248-303 reduced with RCCL).

A one-GPU box cannot show two devices, so the device list names device 0 twice: two contexts, two host threads, two
pipelines, contiguous column ranges, the accumulators of the two contexts added and passed through a (one-rank) RCCL
all-reduce.  The partition arithmetic and the limb conversion are covered on CPU (tests/test_capi_cpu.py)."""
from fractions import Fraction

import numpy as np
import pytest
import torch

import cases
from parity import OUT

pytestmark = pytest.mark.gpu


def _exact_sums(ppt):
    """sum over columns of ppt[:, s] on the library's fixed-point grid (2**-128, magnitudes truncated), as rationals."""
    out = []
    for s in range(4):
        tot = 0
        for x in ppt[:, s].tolist():
            f = Fraction(x)
            v = int(abs(f) * 2 ** 128)
            tot += -v if f < 0 else v
        out.append(float(Fraction(tot, 2 ** 128)))
    return np.array(out)


@pytest.mark.parametrize("name,iiwarm,ncol", [("config3", False, 1001), ("config5", False, 2500), ("config2", True, 37)])
def test_device_list_0_0_equals_single_context(name, iiwarm, ncol):
    from kid_amd import ThompsonMP
    from kid_amd.thompson import ThompsonMulti, shard_bounds
    st = getattr(cases, name)(ncol)
    if name == "config2":
        st["qr"] *= np.linspace(0.5, 1.5, ncol)[:, None]
    single = ThompsonMP(iiwarm=iiwarm)
    one = ThompsonMulti([0], iiwarm=iiwarm)
    two = ThompsonMulti([0, 0], iiwarm=iiwarm)
    try:
        a = {k: v.copy() for k, v in st.items()}
        b = {k: v.copy() for k, v in st.items()}
        c = {k: v.copy() for k, v in st.items()}
        pa, pb, pc = np.zeros((ncol, 4)), np.zeros((ncol, 4)), np.zeros((ncol, 4))
        for _ in range(3):                                           # precipitation reaches the ground in config 5
            pa, ra = single.batch_step_host(a, 10.0, ppt=pa, want_rates=True)
            pb, rb, nb, sb = one.batch_step_host(b, 10.0, ppt=pb, want_rates=True, want_nstep=True)
            pc, rc, nc_, sc = two.batch_step_host(c, 10.0, ppt=pc, want_rates=True, want_nstep=True)
        for k in OUT:
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], c[k]), k     # bitwise per column
        assert np.array_equal(pa, pb) and np.array_equal(pa, pc)
        assert np.array_equal(ra, rb) and np.array_equal(ra, rc)
        assert np.array_equal(nb, nc_)
        # the domain sums: identical bits for both device lists, equal to the exact rational sum, close to a float sum
        assert np.array_equal(sb, sc), (sb, sc)
        want = _exact_sums(pa)
        assert np.all((sc == want) | (np.abs(sc - want) <= 2.3e-16 * np.abs(want))), (sc, want)
        np.testing.assert_allclose(sc, pa.sum(axis=0), rtol=1e-13, atol=0)
        if name == "config5":
            assert sc[0] > 0                                          # rain on the ground: the sums are not trivially zero
        # and the device-side exact reduction a device-resident caller uses gives the same limbs -> same sums
        from kid_amd.thompson import limbs_to_sums
        limbs = single.reduce_ppt_exact(torch.from_numpy(pa).cuda())
        assert np.array_equal(limbs_to_sums(limbs.cpu().numpy()), sc)
        lo, hi = shard_bounds(ncol, 2, 1)
        assert lo == (ncol + 1) // 2 and hi == ncol
    finally:
        single.close(); one.close(); two.close()


def test_config4_eight_way_partition_on_one_device():
    """BASELINE configs[3] as the library's multi entry sees it: a 125 000-column shard's worth of config-3-type columns
    cut eight ways (kidmp_init_multi with the device list (0,)*8: eight contexts, eight host threads, eight pipelines,
    ranges of 15 625 columns) against ONE context: concatenation equality per column, identical exact sums, identical
    sanity scan."""
    from kid_amd import ThompsonMP
    from kid_amd.thompson import ThompsonMulti, limbs_to_sums, shard_bounds
    ncol = 125000
    st = cases.config3(ncol, seed=cases.SEED + 5)
    a = {k: v.copy() for k, v in st.items()}
    single = ThompsonMP(iiwarm=False)
    try:
        pa, _ = single.batch_step_host(a, 10.0)
        limbs = single.reduce_ppt_exact(torch.from_numpy(pa).cuda())
        want_sums = limbs_to_sums(limbs.cpu().numpy())
        dev = {k: torch.from_numpy(a[k]).cuda() for k in single.SANITY_NEG}
        want_sanity = single.sanity(dev).cpu().numpy()
        del dev
    finally:
        single.close()
    eight = ThompsonMulti([0] * 8, iiwarm=False)
    try:
        b = {k: v.copy() for k, v in st.items()}
        pb, _, nb, sums, sanity = eight.batch_step_host(b, 10.0, want_nstep=True, want_sanity=True)
        for k in OUT:
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(pa, pb)
        assert np.array_equal(sums, want_sums), (sums, want_sums)
        assert np.array_equal(sanity, want_sanity), (sanity, want_sanity)
        assert (nb[:, 0] >= 1).all()
        sizes = [shard_bounds(ncol, 8, i)[1] - shard_bounds(ncol, 8, i)[0] for i in range(8)]
        assert sizes == [15625] * 8
    finally:
        eight.close()


@pytest.mark.parametrize("name,iiwarm,ncol", [("config3", False, 3001), ("config2", True, 64)])
def test_multi_sanity_scan_equals_unsharded_scan(name, iiwarm, ncol):
    """kidmp_batch_step_host_multi_diag's sanity15: reduced over the contexts (max of the seven maxima, sum of the eight
    negative counts) it equals kidmp_sanity_device of the unsharded end state, also with negative entries present."""
    from kid_amd import ThompsonMP
    from kid_amd.thompson import ThompsonMulti
    st = getattr(cases, name)(ncol)
    st["qr"][5, 3] = -1e-9            # the scan counts negative entries of the END state: block B zeroes these (q <= R1) ...
    st["qv"][7:9, 100] = -1e-3        # ... and clamps qv to 1e-10, so the counts come out 0 -- the maxima are what differs per shard
    single, two = ThompsonMP(iiwarm=iiwarm), ThompsonMulti([0, 0], iiwarm=iiwarm)
    try:
        a = {k: v.copy() for k, v in st.items()}
        b = {k: v.copy() for k, v in st.items()}
        single.batch_step_host(a, 10.0)
        _, _, _, sums, sanity = two.batch_step_host(b, 10.0, want_sanity=True)
        dev = {k: torch.from_numpy(a[k]).cuda() for k in single.SANITY_NEG}
        want = single.sanity(dev).cpu().numpy()
        assert np.array_equal(sanity, want), (sanity, want)
        for i, k in enumerate(single.SANITY_MAX):
            assert sanity[i] == a[k].max(), k
        # the maxima of the two halves differ (perturbed columns), so the MAX reduction is not trivially one shard's value
        if name == "config3":
            half = ncol // 2 + ncol % 2
            assert a["qr"][:half].max() != a["qr"][half:].max()
        # a second call without the scan leaves no stale state behind and still returns the sums
        c = {k: v.copy() for k, v in st.items()}
        _, _, _, sums2 = two.batch_step_host(c, 10.0)
        assert np.array_equal(sums, sums2)
    finally:
        single.close(); two.close()


def test_reserve_is_a_checked_no_op():
    from kid_amd import KidmpError, ThompsonMP
    m = ThompsonMP(iiwarm=True)
    try:
        m.reserve(1000, 120)
        with pytest.raises(KidmpError):
            m.reserve(-1, 120)
    finally:
        m.close()


def test_multi_entry_optional_arrays_and_errors():
    """The KiD adapter's call (arrays KiD never fills left out) through the multi entry; bad calls fail with a message."""
    from kid_amd import KidmpError, ThompsonMP
    from kid_amd.thompson import ThompsonMulti
    ncol = 300
    st = cases.config2(ncol)
    st["qr"] *= np.linspace(0.5, 1.5, ncol)[:, None]
    keep = ("qv", "qc", "qr", "nr", "t", "p", "dz")
    single, two = ThompsonMP(iiwarm=True), ThompsonMulti([0, 0], iiwarm=True)
    try:
        a = {k: st[k].copy() for k in keep}
        b = {k: st[k].copy() for k in keep}
        pa, _ = single.batch_step_host(a, 10.0)
        pb, _, _, sums = two.batch_step_host(b, 10.0)
        for k in keep:
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(pa, pb)
        # an empty batch is fine and sums to zero
        e = {k: np.zeros((0, 120)) for k in keep}
        _, _, _, s0 = two.batch_step_host(e, 10.0)
        assert np.array_equal(s0, np.zeros(4))
        # fewer columns than contexts: the second range is empty
        c = {k: st[k][:1].copy() for k in keep}
        d = {k: st[k][:1].copy() for k in keep}
        single.batch_step_host(c, 10.0)
        two.batch_step_host(d, 10.0)
        for k in keep:
            assert np.array_equal(c[k], d[k]), k
        bad = {k: st[k].copy() for k in keep if k != "t"}
        bad["t"] = None
        with pytest.raises(KidmpError) as err:
            two.batch_step_host(bad, 10.0)
        assert "null array" in str(err.value) or "device" in str(err.value)
    finally:
        single.close(); two.close()
    with pytest.raises(KidmpError):
        ThompsonMulti([0, 99], iiwarm=True)                          # no such device: refused, nothing leaks
