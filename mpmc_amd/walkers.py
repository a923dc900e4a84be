"""Walker averaging across ranks (one independent Markov chain per GPU).

The reference gathers every rank's observables_t on rank 0 each `corrtime` steps and folds them
into running averages there (src/mc/mc.c:417-476, MPI_Gather + update_root_averages).  Here every
rank contributes a short vector of per-interval sums and one all-reduce gives every rank the pooled
sums, from which mean and standard error follow.  The collective itself is a *reducer*:

  AbiReducer    the production path: walkers_pool_begin/_end of the C host layer, i.e.
                mpmc_hip_allreduce_observables_begin/_end of the C ABI (RCCL over xGMI) -- what the
                reference's C mc() would call in place of MPI_Gather;
  TorchReducer  torch.distributed (gloo on CPU in the tests);
  None          a single walker.

Rank r's chain is seeded with seed + r (the reference's preset_seeds gives every MPI rank the SAME
seed, src/mersenne/mersenne.cpp:13-14: that would make the walkers identical) -- walker_seed().
"""
import numpy as np

FIELDS = ("count", "energy", "energy_sq", "rd_energy", "coulombic_energy", "polarization_energy",
          "polar_iterations", "accepted")


def walker_seed(seed, rank, walkers_per_rank=1, w=0):
    """Seed of walker w of rank `rank`: distinct for every walker of the job."""
    return int(seed) + int(rank) * int(walkers_per_rank) + int(w)


class AbiReducer:
    """Sum over walkers through the C ABI's RCCL entry, held by a HostSystem (mpmc_amd/host.py)."""

    def __init__(self, host_system):
        self.h = host_system

    def begin(self, v):
        self.h.pool_begin(v)

    def end(self, count):
        return self.h.pool_end(count)


class TorchReducer:
    """Sum over ranks with torch.distributed (async all-reduce); used with gloo on CPU."""

    def __init__(self, dist, device=None):
        self.dist = dist
        self.device = device or "cpu"
        self._bufs = None
        self._k = 0
        self._pending = None

    def begin(self, v):
        import torch

        if self._bufs is None:
            self._bufs = [torch.zeros(len(v), dtype=torch.float64, device=self.device) for _ in range(2)]
        buf = self._bufs[self._k & 1]
        self._k += 1
        buf.copy_(torch.from_numpy(np.ascontiguousarray(v)))
        self._pending = (self.dist.all_reduce(buf, async_op=True), buf)

    def end(self, count):
        work, buf = self._pending
        work.wait()
        self._pending = None
        return buf.cpu().numpy()[:count].copy()


class WalkerAverages:
    def __init__(self, reducer=None, dist=None, device=None):
        if reducer is None and dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            reducer = TorchReducer(dist, device)
        self.reducer = reducer
        self.local = np.zeros(len(FIELDS))
        self.pooled = np.zeros(len(FIELDS))
        self.own = np.zeros(len(FIELDS))  # this walker's own totals (never reduced): tells copies of a walker apart
        self._in_flight = False
        self.reductions = 0

    def add(self, energy, rd, es, pol, iters, accepted):
        self.local += (1.0, energy, energy * energy, rd, es, pol, iters, accepted)

    def _finish_pending(self):
        if self._in_flight:
            self.pooled += self.reducer.end(len(FIELDS))
            self._in_flight = False

    def reduce(self):
        """Sum the interval's local sums over all walkers.  The collective is started here and folded into
        the pooled totals at the NEXT call (or at summary()): the averages are only reported, never fed back
        into the chains, so the energy() calls of the next interval hide its latency (the reference blocks
        in MPI_Gather here, mc.c:431)."""
        v = self.local.copy()
        self.local[:] = 0.0
        self.own += v
        if self.reducer is not None:
            self._finish_pending()
            self.reducer.begin(v)
            self._in_flight = True
            self.reductions += 1
        else:
            self.pooled += v

    def own_mean_energy(self):
        return float(self.own[1] / max(self.own[0], 1.0))

    def summary(self):
        self._finish_pending()
        n = max(self.pooled[0], 1.0)
        mean = self.pooled[1] / n
        var = max(self.pooled[2] / n - mean * mean, 0.0)
        return dict(samples=int(self.pooled[0]), energy=mean, energy_sdom=(var / n) ** 0.5,
                    rd_energy=self.pooled[3] / n, coulombic_energy=self.pooled[4] / n,
                    polarization_energy=self.pooled[5] / n, polar_iterations=self.pooled[6] / n,
                    acceptance=self.pooled[7] / n)
