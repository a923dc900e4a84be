"""A few FULL (non-incremental) evaluations of the headline workload: what the VALU-bound kernels cost over all tiles
(pair_rd_es_kernel<4>: LJ + real-space Ewald; static_field_kernel<0>: Thole static field).  Run under rocprofv3 --pmc by
profiles/valu_counters.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpmc_amd import engine

s = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pcn61_bssp_4096.npz")))
p = dict(temperature=77.0, polarization=1, polar_damp=2.1304, polar_max_iter=4, pbc_cutoff=8.0, feynman_hibbs=1, feynman_hibbs_order=4)
e = engine.Engine(len(s["charge"]))
e.load_system(s, p)
e.set_option("incremental_pairs", 0)
e.set_option("fuse_recip", 0)
pos = s["pos"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for k in range(n):
    a = 2016 + 5 * k
    e.update_atoms(a, pos[a:a + 5] + 0.01 * (k + 1))
    e.energy()
e.close()
