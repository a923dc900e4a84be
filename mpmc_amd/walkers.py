"""Walker averaging across ranks (one independent Markov chain per GPU).

The reference gathers every rank's observables_t on rank 0 each `corrtime` steps and folds them
into running averages there (src/mc/mc.c:417-476, MPI_Gather + update_root_averages).  Here every
rank contributes a short vector of per-interval sums; one all-reduce (RCCL over xGMI on GPUs, gloo
on CPU in the tests) gives every rank the pooled sums, from which mean and standard error follow.
"""
import numpy as np

FIELDS = ("count", "energy", "energy_sq", "rd_energy", "coulombic_energy", "polarization_energy",
          "polar_iterations", "accepted")


class WalkerAverages:
    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.device = device
        self.local = np.zeros(len(FIELDS))
        self.pooled = np.zeros(len(FIELDS))
        self._bufs = None
        self._pending = None  # (work handle, buffer) of the all-reduce still in flight
        self._k = 0

    def add(self, energy, rd, es, pol, iters, accepted):
        self.local += (1.0, energy, energy * energy, rd, es, pol, iters, accepted)

    def _distributed(self):
        return self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1

    def _finish_pending(self):
        if self._pending is not None:
            work, buf = self._pending
            work.wait()
            self.pooled += buf.cpu().numpy()
            self._pending = None

    def reduce(self):
        """Sum the interval's local sums over all walkers.  The collective is launched asynchronously and
        folded into the pooled totals at the NEXT call (or at summary()): the averages are only reported,
        never fed back into the chains, so the ~10 energy() calls of the next interval hide its latency
        (the reference blocks in MPI_Gather here, mc.c:431)."""
        v = self.local.copy()
        self.local[:] = 0.0
        if self._distributed():
            import torch

            if self._bufs is None:
                self._bufs = [torch.zeros(len(FIELDS), dtype=torch.float64, device=self.device or "cpu")
                              for _ in range(2)]
            self._finish_pending()
            buf = self._bufs[self._k & 1]
            self._k += 1
            buf.copy_(torch.from_numpy(v))
            self._pending = (self.dist.all_reduce(buf, async_op=True), buf)
        else:
            self.pooled += v

    def summary(self):
        self._finish_pending()
        n = max(self.pooled[0], 1.0)
        mean = self.pooled[1] / n
        var = max(self.pooled[2] / n - mean * mean, 0.0)
        return dict(samples=int(self.pooled[0]), energy=mean, energy_sdom=(var / n) ** 0.5,
                    rd_energy=self.pooled[3] / n, coulombic_energy=self.pooled[4] / n,
                    polarization_energy=self.pooled[5] / n, polar_iterations=self.pooled[6] / n,
                    acceptance=self.pooled[7] / n)
