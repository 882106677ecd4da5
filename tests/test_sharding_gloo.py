"""Multi-rank path on CPU: world_size-2 gloo.  Columns shard with no data-path collective;
the only exchange is the all-reduce of the domain diagnostics.  Without a GPU the stepper of the
first test is the oracle (allowed in tests) and the second test drives bench.py's own launcher in its
rehearsal mode (no physics); on the GPU box tests/test_gpu_sharding.py runs the same launcher and
kid_amd.sharding.ShardedColumns on the HIP path."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch.multiprocessing as mp

import cases
from kid_amd.sharding import shard_bounds


def test_shard_bounds_partition():
    for ncol in (1, 7, 10000, 100003):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(ncol, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == ncol
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    from kid_amd.sharding import allreduce_precip_sums, shard_state
    from oracle.oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = cases.replicate(cases.warm_column_t900(), 10)
    st["qr"] *= np.linspace(0.5, 1.5, 10)[:, None]          # make the columns differ
    mine = {k: np.ascontiguousarray(v) for k, v in shard_state(st, rank, world).items()}
    o = Oracle(iiwarm=True, nthreads=1)
    tot = np.zeros(4)
    for _ in range(100):                                    # until rain reaches the surface
        tot += o.batch_step(mine, 10.0).sum(axis=0)
    sums = torch.from_numpy(tot.copy())
    allreduce_precip_sums(sums)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), sums=sums.numpy(), **mine)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_run_equals_single_rank(tmp_path, oracle_warm):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    st = cases.replicate(cases.warm_column_t900(), 10)
    st["qr"] *= np.linspace(0.5, 1.5, 10)[:, None]
    tot = np.zeros(4)
    for _ in range(100):
        tot += oracle_warm.batch_step(st, 10.0).sum(axis=0)
    assert tot[0] > 0
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for k in ("qv", "qc", "qr", "nr", "t"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), st[k]), k     # concatenation equality
    for p in parts:
        np.testing.assert_allclose(p["sums"], tot, rtol=1e-13)                     # all-reduced diagnostics


def test_bench_launcher_spawns_ranks_and_reduces():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent must start two fresh rank processes with
    RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set, they must rendezvous on 127.0.0.1 and all-reduce, and rank 0 must
    print ONE line with n_gpus = 2.  CPU box: --rehearse-launcher (no physics, value 0)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--rehearse-launcher", "--steps", "4", "--warmup", "1", "--ncol", "10"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] == 0.0 and "no physics" in d["data"]
    # rank r adds 1e-3*dt*(r+1) per column and step (5 steps incl. warm-up): the reduced sum proves both ranks ran
    np.testing.assert_allclose(d["precip_domain_sums"][0], 10 * 5 * 1e-3 * 10.0 * (1 + 2), rtol=1e-12)


def test_bench_under_torch_distributed_run():
    """The driver's own form at N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- the ranks exist already (WORLD_SIZE is set), so bench.py must
    not spawn any and rank 0 must print the one line."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--backend", "gloo", "--rehearse-launcher", "--steps", "3", "--warmup", "1", "--ncol", "10"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    np.testing.assert_allclose(d["precip_domain_sums"][0], 10 * 4 * 1e-3 * 10.0 * (1 + 2), rtol=1e-12)


def test_bench_refuses_to_fake_a_gpu_run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
