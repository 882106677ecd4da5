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


class ShardedColumns:
    """One rank's share of a column batch: the product-side object behind `bench.py --gpus N` and the
    multi-GPU tests.  Owns a ThompsonMP context bound to this rank's GPU, the shard's state in HBM and its
    precipitation accumulators; `step` advances the shard (no communication), `diagnostics` reduces the
    domain diagnostics over all ranks -- the one exchange of the path."""

    def __init__(self, state, rank, world, device, iiwarm, set_Nc=100.0, l_sediment=True, want_rates=False,
                 local=False, arith="p64"):
        """state: dict of [ncol, nz] float64 arrays (numpy or torch) -- the GLOBAL batch, of which this rank takes its
        contiguous range; or, with local=True, this rank's own columns (weak-scaling runs generate them per rank)."""
        import torch
        from .thompson import NRATES, ThompsonMP
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        n = next(iter(state.values())).shape[0]
        if local:
            self.lo, self.hi, self.ncol_global = 0, n, n * self.world
        else:
            self.lo, self.hi = shard_bounds(n, rank, world)
            self.ncol_global = n
        self.model = ThompsonMP(iiwarm=iiwarm, set_Nc=set_Nc, l_sediment=l_sediment, device=self.device)
        dev = torch.device("cuda", self.device)
        # arith: "p64" (binary64 state, the parity build) or the binary32-state builds "p32n" / "f32" (kidmp32_*)
        self.arith = arith
        dt_state = torch.float64 if arith == "p64" else torch.float32
        self.st = {k: torch.as_tensor(v[self.lo:self.hi]).contiguous().to(dev).to(dt_state) for k, v in state.items()}
        self.ncol, self.nz = self.st["qv"].shape
        self.ppt = torch.zeros(self.ncol, 4, dtype=dt_state, device=dev)
        self.rates = torch.zeros(self.ncol, NRATES, self.nz, dtype=torch.float64, device=dev) if want_rates else None

    def step(self, dt):
        """One mp_thompson advance of every column of the shard, asynchronous on torch's current stream of the
        shard's device."""
        if self.arith == "p64":
            self.model.batch_step(self.st, dt, self.ppt, rates=self.rates)
        else:
            self.model.batch_step32(self.st, dt, self.ppt, arith=self.arith, rates=self.rates)

    def synchronize(self):
        import torch
        torch.cuda.synchronize(self.device)

    def diagnostics(self, group=None, cpu_collective=False, want_sanity=True):
        """Domain diagnostics over ALL ranks: dict(precip=[4] sums, precip_limbs=int64 [24] exact accumulators they
        come from, sanity=[15] (7 maxima, 8 negative counts), rates=[36, nz] sums or None).  One RCCL all-gather of the per-rank vectors, reduced locally (SUM, and MAX for
        the maxima); cpu_collective=True moves them to the host first (gloo rehearsals).  want_sanity=False leaves the
        max-q / negative-value scan out (the optional debugging aid of SURVEY 8e; it reads eight state arrays) and
        returns zeros in its place: what remains is the reference adapter's own exchange, the precipitation means of
        W:248-303."""
        import torch
        import torch.distributed as dist
        from .thompson import PPT_LIMBS, limbs_to_sums
        # precipitation: exact fixed-point accumulators (int64 limbs) -- integer sums are associative, so the reduced
        # domain sums are the same bits for every world size and every partition of the columns
        ppt64 = self.ppt if self.arith == "p64" else self.ppt.double()        # the diagnostics kernels are binary64
        limbs = self.model.reduce_ppt_exact(ppt64)
        if want_sanity:
            sanity = self.model.sanity(self.st if self.arith == "p64" else {k: self.st[k].double() for k in self.model.SANITY_NEG})
        else:
            sanity = torch.zeros(15, dtype=torch.float64, device=limbs.device)
        rates = self.model.reduce_rates(self.rates) if self.rates is not None else None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            # ONE collective: every rank contributes its [24 limbs (bit patterns) + 15 (+ 36 nz)] vector; the sums and
            # maxima are then formed locally in rank order (identical on all ranks).  <= 35 KB per rank: latency-bound
            # whatever the links.  An all-gather moves bits, it does no arithmetic: the int64 limbs ride along as the
            # bit patterns of float64 slots.
            world = dist.get_world_size(group)
            parts = [limbs.view(torch.float64), sanity] + ([rates.reshape(-1)] if rates is not None else [])
            mine = torch.cat(parts)
            if cpu_collective:
                mine = mine.cpu()
            gathered = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
            with torch.cuda.device(self.device):
                dist.all_gather_into_tensor(gathered, mine, group=group)
            g = gathered.view(world, -1)
            limbs = g[:, 0:PPT_LIMBS].contiguous().view(torch.int64).sum(dim=0)          # wrap-around int64 sum
            o = PPT_LIMBS
            sanity = torch.cat([g[:, o:o + 7].max(dim=0).values, g[:, o + 7:o + 15].sum(dim=0)])
            if rates is not None:
                rates = g[:, o + 15:].sum(dim=0).view(rates.shape)
        precip = torch.from_numpy(limbs_to_sums(limbs.cpu().numpy())).to(sanity.device)
        return dict(precip=precip, precip_limbs=limbs, sanity=sanity, rates=rates)

    def close(self):
        self.model.close()
