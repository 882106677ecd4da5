"""Multi-rank path on CPU: world_size-2 gloo.  Columns shard with no data-path collective;
the only exchange is the all-reduce of the 4 precipitation sums.  The stepper here is the
oracle (allowed in tests); on GPUs the same sharding code feeds ThompsonMP.batch_step."""
import os
import socket

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
