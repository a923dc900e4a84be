"""The N > 1 path on CPU: two ranks (gloo), each an independent walker, pooled through the same
WalkerAverages.reduce() that bench.py runs over RCCL on the GPUs."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from mpmc_amd.walkers import WalkerAverages
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(100 + rank)          # seed + rank, as the walkers do
    avg = WalkerAverages(dist=dist)
    mine = []
    for step in range(1, 41):
        e = -1000.0 + rng.normal()
        mine.append(e)
        avg.add(e, 0.5 * e, 0.3 * e, 0.2 * e, 4, step %% 2)
        if step %% 10 == 0:
            avg.reduce()
    s = avg.summary()
    # every rank must hold the same pooled result
    gathered = [None] * world
    dist.all_gather_object(gathered, (s, mine))
    if rank == 0:
        allv = np.concatenate([np.array(g[1]) for g in gathered])
        assert all(abs(g[0]["energy"] - s["energy"]) == 0.0 for g in gathered)
        assert s["samples"] == 40 * world
        assert abs(s["energy"] - allv.mean()) < 1e-9
        assert abs(s["energy_sdom"] - allv.std() / np.sqrt(len(allv))) < 1e-9
        assert abs(s["polar_iterations"] - 4.0) < 1e-12 and abs(s["acceptance"] - 0.5) < 1e-12
        print("OK", json.dumps(s))
    dist.destroy_process_group()
""") % ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_walkers_pool_observables_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


def test_single_walker_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from mpmc_amd.walkers import WalkerAverages

    a = WalkerAverages()
    a.add(-10.0, -5.0, -3.0, -2.0, 4, 1)
    a.add(-12.0, -6.0, -4.0, -2.0, 4, 0)
    a.reduce()
    s = a.summary()
    assert s["samples"] == 2 and s["energy"] == -11.0 and s["acceptance"] == 0.5
