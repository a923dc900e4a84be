"""CPU-only checks of the drop-in boundary: the library loads and exports every symbol that
include/mpmc_hip.h declares; struct layouts agree between the header and the ctypes mirror."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "mpmc_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from mpmc_amd import engine

    if not os.path.exists(engine.LIB_PATH):
        g.build()
    return engine.load()


def header_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpmc_hip_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from mpmc_amd import engine

    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(engine.EXPORTS) == names


def test_abi_version(lib):
    assert lib.mpmc_hip_abi_version() == 1


def test_struct_layouts_match_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    from mpmc_amd import engine

    prog = tmp_path / "layout.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mpmc_hip.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu\\n\", sizeof(mpmc_hip_params), sizeof(mpmc_hip_result),"
        " sizeof(mpmc_hip_timings), offsetof(mpmc_hip_params, polar_wolf_alpha),"
        " offsetof(mpmc_hip_result, polar_iterations), offsetof(mpmc_hip_timings, sweep_count));return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(engine.Params), C.sizeof(engine.Result), C.sizeof(engine.Timings),
            engine.Params.polar_wolf_alpha.offset, engine.Result.polar_iterations.offset,
            engine.Timings.sweep_count.offset]
    assert got == want


def test_default_params(lib):
    from mpmc_amd import engine

    p = engine.make_params()
    assert p.rd_lrc == 1 and p.ewald_kmax == 7 and p.polar_max_iter == 10 and p.polar_gamma == 1.0


def test_no_device_fails_loudly(lib):
    """Without a GPU the engine must refuse to create a context (no CPU fallback)."""
    from mpmc_amd import engine

    if lib.mpmc_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(engine.EngineError):
        engine.Engine(16)
