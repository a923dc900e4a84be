"""CPU-only tests of the C host layer (mpmc_amd/host/): RNG stream, config + PQR readers, moves, list flattening.
energy() itself needs the GPU and is covered in test_gpu_host.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mpmc_amd import host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g

    if not os.path.exists(host.LIB_PATH):
        g.build()


def test_get_rand_is_std_mt19937_uniform_real(tmp_path):
    """get_rand() must reproduce std::mt19937 + std::uniform_real_distribution<double>(0,1)
    (reference src/mersenne/mersenne.cpp:9-22) draw for draw."""
    src = tmp_path / "r.cpp"
    src.write_text('#include <random>\n#include <cstdio>\nint main(){std::mt19937 rng; rng.seed(752498u);'
                   'std::uniform_real_distribution<double> uid(0.0,1.0);'
                   'for(int i=0;i<2000;i++) printf("%.17g\\n", uid(rng)); return 0;}\n')
    exe = tmp_path / "r"
    subprocess.check_call(["g++", "-O1", str(src), "-o", str(exe)])
    want = np.array([float(x) for x in subprocess.check_output([str(exe)]).split()])
    h = host.HostSystem(synth.s_pol(10), synth.FLAGS_POL_JACOBI, seed=752498)
    got = np.array([h.lib.host_get_rand(h.ptr) for _ in range(2000)])
    h.close()
    assert np.array_equal(got, want)


def test_setup_system_reads_reference_style_inputs():
    lib = host.load()
    p = lib.setup_system(os.path.join(ROOT, "tests", "data", "bssp_small", "input").encode())
    assert p
    assert lib.countNatoms(p) == 10
    pos = np.zeros((10, 3))
    lib.host_get_positions(p, pos.ctypes.data)
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "bssp_small_10.npz")))
    assert np.allclose(pos, gold["pos"], atol=1e-12)
    lib.free_system(p)


def test_bad_keyword_is_rejected(tmp_path):
    (tmp_path / "in").write_text("ensemble nvt\nnot_a_keyword 3\n")
    assert not host.load().setup_system(str(tmp_path / "in").encode())


def test_precision_and_max_iter_conflict_is_rejected(tmp_path):
    """reference check_input.c:424-428"""
    gold = os.path.join(ROOT, "tests", "data", "bssp_small", "small.initial.pqr")
    (tmp_path / "in").write_text(
        "ensemble nvt\ntemperature 77\npolarization on\npolar_damp_type exponential\npolar_damp 2.1304\n"
        "polar_iterative on\npolar_precision 1e-5\nbasis1 20 0 0\nbasis2 0 20 0\nbasis3 0 0 20\npqr_input %s\n" % gold)
    assert not host.load().setup_system(str(tmp_path / "in").encode())


def test_moves_are_rigid_and_restore_is_exact():
    """translate + quaternion rotate (reference mc_moves.c:378-488) move exactly one molecule rigidly;
    restore() (mc_moves.c:744-807) puts it back bit for bit."""
    s = synth.s_pol(50)
    h = host.HostSystem(s, synth.FLAGS_POL_JACOBI, seed=7, move_factor=0.1, rot_factor=0.2)
    lib = h.lib
    lib.host_init_chain_no_energy.argtypes = [C.c_void_p]
    lib.host_init_chain_no_energy(h.ptr)
    before = h.positions()
    for _ in range(20):
        lib.make_move(h.ptr)
        after = h.positions()
        moved = np.flatnonzero(np.any(after != before, axis=1))
        assert 0 < len(moved) <= 5 and len(set(s["molecule"][moved])) == 1
        a, b = moved.min(), moved.max() + 1
        d0 = np.linalg.norm(before[a:b, None] - before[None, a:b], axis=-1)
        d1 = np.linalg.norm(after[a:b, None] - after[None, a:b], axis=-1)
        assert np.allclose(d0, d1, atol=1e-12)
        lib.restore(h.ptr)
        assert np.array_equal(h.positions(), before)
    h.close()


def test_flatten_order_and_config_text():
    txt = host.config_text(synth.FLAGS_POL_PRODUCTION)
    assert "polar_gs_ranked on" in txt and "polar_wolf_alpha 0.13" in txt and "polar_damp_type exponential" in txt
    s = synth.s_pol(25)
    h = host.HostSystem(s, synth.FLAGS_POL_PRODUCTION)
    assert np.array_equal(h.positions(), s["pos"])
    h.close()
