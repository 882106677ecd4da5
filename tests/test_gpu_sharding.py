"""Multi-GPU path on PRODUCT code (-m gpu): kid_amd.sharding.ShardedColumns over ThompsonMP contexts.

A one-GPU box cannot host two RCCL ranks, so the sharding itself (contiguous column ranges, independent contexts,
reduced diagnostics) is checked with two contexts on cuda:0 -- the launcher/rendezvous half is covered on CPU by
tests/test_sharding_gloo.py and, with both halves together, by `bench.py --gpus 2 --backend gloo` below."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import cases
from parity import OUT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,iiwarm,ncol", [("config3", False, 1001), ("config5", False, 514), ("config2", True, 37)])
def test_two_shards_equal_one_batch(name, iiwarm, ncol):
    """shard_state over two contexts: concatenation equality (bitwise) and reduced sums vs the unsharded run."""
    from kid_amd.sharding import ShardedColumns
    st = getattr(cases, name)(ncol)
    if name == "config2":
        st["qr"] *= np.linspace(0.5, 1.5, ncol)[:, None]
    whole = ShardedColumns(st, 0, 1, 0, iiwarm, want_rates=True)
    parts = [ShardedColumns(st, r, 2, 0, iiwarm, want_rates=True) for r in range(2)]
    try:
        assert parts[0].hi == parts[1].lo and parts[0].lo == 0 and parts[1].hi == ncol
        for _ in range(3):
            whole.step(10.0)
            for p in parts:
                p.step(10.0)
        torch.cuda.synchronize()
        for k in OUT:
            assert torch.equal(torch.cat([p.st[k] for p in parts]), whole.st[k]), k
        assert torch.equal(torch.cat([p.ppt for p in parts]), whole.ppt)
        assert torch.equal(torch.cat([p.rates for p in parts]), whole.rates)
        dw = whole.diagnostics()
        dp = [p.diagnostics() for p in parts]
        # the reductions a multi-rank run all-reduces: SUM of the precipitation and rate sums, MAX / SUM of the scan
        np.testing.assert_allclose((dp[0]["precip"] + dp[1]["precip"]).cpu().numpy(), dw["precip"].cpu().numpy(), rtol=1e-13)
        # the exchange itself is exact: the int64 accumulators of the shards add up to the unsharded ones bit for bit,
        # hence so do the domain sums formed from them (what a multi-rank run all-reduces)
        from kid_amd.thompson import limbs_to_sums
        both = dp[0]["precip_limbs"] + dp[1]["precip_limbs"]
        assert torch.equal(both, dw["precip_limbs"])
        assert np.array_equal(limbs_to_sums(both.cpu().numpy()), dw["precip"].cpu().numpy())
        np.testing.assert_allclose((dp[0]["rates"] + dp[1]["rates"]).cpu().numpy(), dw["rates"].cpu().numpy(),
                                   rtol=1e-12, atol=1e-30)
        assert torch.equal(torch.maximum(dp[0]["sanity"][:7], dp[1]["sanity"][:7]), dw["sanity"][:7])
        assert torch.equal(dp[0]["sanity"][7:] + dp[1]["sanity"][7:], dw["sanity"][7:])
        # and against plain torch reductions of the same device data
        np.testing.assert_allclose(dw["precip"].cpu().numpy(), whole.ppt.sum(dim=0).cpu().numpy(), rtol=1e-12)
        np.testing.assert_allclose(dw["rates"].cpu().numpy(), whole.rates.sum(dim=0).cpu().numpy(), rtol=1e-11, atol=1e-30)
        for i, k in enumerate(("qc", "qr", "nr", "qs", "qi", "qg", "ni")):
            assert float(dw["sanity"][i]) == float(whole.st[k].max()), k
        assert float(dw["sanity"][7:].sum()) == 0.0                      # no negative mixing ratios or numbers
    finally:
        whole.close()
        for p in parts:
            p.close()


def test_sanity_scan_counts_negatives(gpu_mixed):
    st = {k: torch.from_numpy(v).cuda() for k, v in cases.config3(8).items()}
    st["qr"][3, 7] = -1e-9
    st["qv"][0, 0] = -1e-5
    st["qv"][5, 119] = -2e-5
    out = gpu_mixed.sanity(st).cpu().numpy()
    assert out[7:].tolist() == [0, 1, 0, 0, 0, 0, 0, 2]
    assert out[1] == float(st["qr"].max())


def test_bench_launcher_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` as the driver starts it at N > 1 without torchrun: the parent spawns the ranks
    (before touching the GPU), they rendezvous on 127.0.0.1, step their shards on the card and all-reduce the
    diagnostics.  gloo because one card cannot host two RCCL ranks; on a multi-GPU node the default backend is nccl."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--ncol", "2000", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["ncol_per_gpu"] == 2000
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--ncol", "2000",
                          "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    # identical replicated columns on both ranks: the all-reduced precipitation is twice one rank's
    np.testing.assert_allclose(d["precip_domain_sums"], 2 * np.asarray(d1["precip_domain_sums"]), rtol=1e-12)


def test_context_survives_other_current_device_and_rejects_foreign_tensors(gpu_mixed):
    """Device guard: the context launches on ITS device whatever torch's current device is, and restores the
    caller's; tensors of another device are refused.  (On a one-GPU box only the first half can run.)"""
    from kid_amd import KidmpError
    st = cases.config3(16)
    dev = {k: torch.from_numpy(v).cuda() for k, v in st.items()}
    ppt = torch.zeros(16, 4, dtype=torch.float64, device="cuda:0")
    before = torch.cuda.current_device()
    gpu_mixed.batch_step(dev, 10.0, ppt)
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step({k: v.cpu() for k, v in dev.items()}, 10.0, ppt)          # host tensors
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step(dev, 10.0, ppt.to(torch.float32))                          # wrong dtype
    bad_w = dict(dev)
    bad_w["w"] = dev["w"][:, :7]
    with pytest.raises(KidmpError):
        gpu_mixed.batch_step(bad_w, 10.0, ppt)                                          # w is validated too
    if torch.cuda.device_count() > 1:
        from kid_amd import ThompsonMP
        other = ThompsonMP(iiwarm=False, device=1)
        try:
            with pytest.raises(KidmpError):
                other.batch_step(dev, 10.0, ppt)                                        # cuda:0 tensors, cuda:1 context
            d1 = {k: v.to("cuda:1") for k, v in dev.items()}
            p1 = torch.zeros(16, 4, dtype=torch.float64, device="cuda:1")
            torch.cuda.set_device(0)                                                    # current device != context's
            other.batch_step({k: torch.from_numpy(st[k]).to("cuda:1") for k in st}, 10.0, p1)
            torch.cuda.synchronize(1)
            assert torch.cuda.current_device() == 0
        finally:
            other.close()
