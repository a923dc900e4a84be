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
        self._buf = None

    def add(self, energy, rd, es, pol, iters, accepted):
        self.local += (1.0, energy, energy * energy, rd, es, pol, iters, accepted)

    def reduce(self):
        """Sum the interval's local sums over all walkers and fold them into the pooled totals."""
        v = self.local.copy()
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            import torch

            if self._buf is None:
                self._buf = torch.zeros(len(FIELDS), dtype=torch.float64, device=self.device or "cpu")
            self._buf.copy_(torch.from_numpy(v))
            self.dist.all_reduce(self._buf)
            v = self._buf.cpu().numpy()
        self.pooled += v
        self.local[:] = 0.0
        return v

    def summary(self):
        n = max(self.pooled[0], 1.0)
        mean = self.pooled[1] / n
        var = max(self.pooled[2] / n - mean * mean, 0.0)
        return dict(samples=int(self.pooled[0]), energy=mean, energy_sdom=(var / n) ** 0.5,
                    rd_energy=self.pooled[3] / n, coulombic_energy=self.pooled[4] / n,
                    polarization_energy=self.pooled[5] / n, polar_iterations=self.pooled[6] / n,
                    acceptance=self.pooled[7] / n)
