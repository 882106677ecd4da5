"""Column sharding for the multi-GPU path.

Columns are independent (the reference's `do i=1,nx` body touches column i only, W:54-246;
lookup tables are read-only), so an N-GPU run is: contiguous ranges of the column index, one
process per GPU, tables replicated per device, NO data-path collective.  The only exchange is
the reduction of the surface-precipitation diagnostics (the nx-means of W:248-275)."""


def shard_bounds(ncol, rank, world):
    """Contiguous [lo, hi) of the column index owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(int(ncol), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_state(state, rank, world):
    """Slices every [ncol, ...] array of a state dict to this rank's columns (views, no copy)."""
    ncol = next(iter(state.values())).shape[0]
    lo, hi = shard_bounds(ncol, rank, world)
    return {k: v[lo:hi] for k, v in state.items()}


def allreduce_precip_sums(sums4, group=None):
    """Sum the per-rank [4] precipitation sums over all ranks (RCCL on GPUs, gloo in CPU tests).
    This is the only collective of the path; 32 bytes, latency-bound."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums4, op=dist.ReduceOp.SUM, group=group)
    return sums4
