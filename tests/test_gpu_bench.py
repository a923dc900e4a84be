"""The bench.py contract on a real GPU: one JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["value"] > 10.0
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"].startswith("hbm") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.2 < rf["frac"] < 1.0
    assert rf["launches"] >= 4 and rf["launches"] % 4 == 0  # Jacobi x4 per step of the separate probe pass, HIP events
    assert rf["avg_launch_ms"] > 0 and "incremental (bit-identical to full)" in d["config"]["workload"]
    assert d["config"]["rccl_ranks"] == 1 and d["full_rebuild_steps_per_s"] > 10.0
    for k in ("pair_rd_es_kernel", "static_field_kernel"):
        assert d["valu_kernels"][k]["pairs_per_s"] > 1e9
    # the counters that would silently change what the line measures: a dedicated box shows 0 / 0
    assert d["resident_fallbacks"] == 0 and d["spec_rank_redos"] == 0
    assert len(d["ranks"]) == 1 and d["ranks"][0]["rank"] == 0 and d["ranks"][0]["steps_per_s"] > 10.0


def test_bench_under_a_launcher_pools_through_the_c_abi_collective_with_one_rank():
    """Under `torch.distributed.run` the walker pooling goes through the C ABI's RCCL entry also with ONE rank: the
    unique id is made by rank 0 and handed round by the launcher's process group, the communicator is created on the
    engine's device beside torch's own RCCL, and every corrtime an all-reduce is started and collected.  (Two ranks
    need two GPUs: next test.)"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and "RCCL" in d["config"]["collective"]
    assert d["walker_averages"]["samples"] == 5 and d["value"] > 10.0


def test_two_walkers_pool_through_the_c_abi_collective():
    """configs[4] in small: `bench.py --gpus 2` starts its two ranks itself, each walker on its own GPU, observables
    pooled by mpmc_hip_allreduce_observables_begin/_end (RCCL).  Needs two GPUs: skipped on the one-GPU box."""
    sys.path.insert(0, ROOT)
    import bench

    if bench.visible_gpus() < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "5",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["walkers"] == 2
    # one sample per corrtime interval per walker: warm-up 5 steps = 1 interval, 40 steps = 4 intervals
    assert d["walker_averages"]["samples"] == 2 * 5
