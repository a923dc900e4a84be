"""The reference-side binding is compiled against the reference's REAL headers (build container only).

SURVEY section 8(b): the drop-in boundary is `double energy(system_t*)` of smann95/mpmc.  The file a maintainer
copies into the reference tree is mpmc_amd/host/energy_hip.c (the same source the GPU tests run, there built
against this repository's mirror header), and the lines added to the reference's own files are collected in
integration/reference_hooks.c.  Here both are type-checked against /root/reference/src/include/{structs.h,
function_prototypes.h, defines.h}.  The only change those headers need is the keyword flag `int hip;` in
system_t: the test adds that line to a scratch copy of structs.h under a temporary directory (nothing of the
reference is written into this repository).  <mc.h> is not used: it includes cmake_config.h, a file only the
reference's CMake configure step generates.

/root/reference does not exist on the GPU box: everything here skips there.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/src/include"
SHIM = os.path.join(ROOT, "mpmc_amd", "host", "energy_hip.c")
HOOKS = os.path.join(ROOT, "integration", "reference_hooks.c")

needs_reference = pytest.mark.skipif(not os.path.isfile(os.path.join(REF_INC, "structs.h")),
                                     reason="the reference tree is not present (GPU box)")


@pytest.fixture(scope="module")
def patched_include(tmp_path_factory):
    """scratch include dir holding structs.h + the one new member"""
    d = tmp_path_factory.mktemp("ref_inc")
    text = open(os.path.join(REF_INC, "structs.h")).read()
    assert text.count("    int cuda;\n") == 1, "structs.h:350 moved: update INTEGRATION.md section 2"
    (d / "structs.h").write_text(text.replace("    int cuda;\n", "    int cuda;\n    int hip;\n"))
    return str(d)


def gcc(args, **kw):
    return subprocess.run(["gcc", "-std=gnu99"] + args, capture_output=True, text=True, **kw)


def include_flags(patched):
    # the scratch copy first, so that `#include <structs.h>` finds the patched one; defines.h etc. from the reference
    return ["-I", patched, "-I", REF_INC, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "mpmc_amd", "host")]


@needs_reference
@pytest.mark.parametrize("src", [SHIM, HOOKS], ids=["energy_hip.c", "reference_hooks.c"])
def test_compiles_against_the_reference_headers(src, patched_include):
    r = gcc(["-fsyntax-only", "-Wall", "-Wextra", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
             "-Werror=int-conversion", "-Werror=implicit-int", "-Wno-unused-parameter"] + include_flags(patched_include) + [src])
    assert r.returncode == 0, r.stderr
    # no diagnostics at all about members, types or prototypes (warnings of the reference's own headers aside)
    own = [l for l in r.stderr.splitlines() if os.path.basename(src) in l and ("warning" in l or "error" in l)]
    assert not own, "\n".join(own)


@needs_reference
def test_without_the_hip_member_it_does_not_compile():
    """the scratch copy is what makes it compile: against the untouched header `system->hip` is an error, i.e. the
    test really reads the reference's system_t (and `int hip;` is the whole header patch)"""
    r = gcc(["-fsyntax-only", "-I", REF_INC, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "mpmc_amd", "host"), HOOKS])
    assert r.returncode != 0 and "no member named 'hip'" in r.stderr.replace("‘", "'").replace("’", "'")
    # the binding itself reads no engine member of system_t at all: it compiles against the untouched header
    r = gcc(["-fsyntax-only", "-I", REF_INC, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "mpmc_amd", "host"), SHIM])
    assert r.returncode == 0, r.stderr


def _prototypes():
    text = open(os.path.join(REF_INC, "function_prototypes.h")).read()
    return set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))


@needs_reference
def test_every_symbol_the_binding_needs_exists(patched_include, tmp_path):
    """undefined symbols of the compiled binding = C ABI entry points libmpmc_hip.so exports + functions the
    reference declares in function_prototypes.h (+ its `rank` global) + libc / libm"""
    obj = str(tmp_path / "energy_hip.o")
    r = gcc(["-c", "-O1", "-o", obj] + include_flags(patched_include) + [SHIM])
    assert r.returncode == 0, r.stderr
    undefined = set(subprocess.check_output(["nm", "-u", obj], text=True).split()[1::2])
    lib = os.path.join(ROOT, "mpmc_amd", "csrc", "libmpmc_hip.so")
    exported = {l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True).splitlines()}
    abi = {s for s in undefined if s.startswith("mpmc_hip_")}
    assert abi and abi <= exported, sorted(abi - exported)
    libc = {"calloc", "malloc", "realloc", "free", "memcpy", "memset", "memcmp", "snprintf", "fprintf", "getenv", "atoi", "atof",
            "clock_gettime", "nanosleep", "fopen", "fclose", "fread", "fwrite", "rename", "stderr", "__stack_chk_fail",
            "_GLOBAL_OFFSET_TABLE_", "strtol", "strtod"}
    rest = {s for s in undefined - abi - libc if not (s.startswith("__") and s.endswith("_chk"))}  # fortified libc
    declared = _prototypes() | {"rank", "size"}
    assert rest <= declared, sorted(rest - declared)
    # and these are the reference functions it leans on
    assert {"error", "update_com", "countN", "countNatoms", "wrapall"} <= rest


def _snippets(md):
    """```c blocks of INTEGRATION.md that follow an `<!-- excerpt: FILE -->` marker"""
    out = []
    for m in re.finditer(r"<!-- excerpt: ([^ ]+) -->\s*```c\n(.*?)```", md, re.S):
        out.append((m.group(1), m.group(2)))
    return out


def _norm(line):
    return re.sub(r"\s+", " ", line.strip())


def test_integration_md_snippets_are_excerpts_of_the_checked_files():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    snippets = _snippets(md)
    assert len(snippets) >= 6
    for fname, body in snippets:
        src = {_norm(l) for l in open(os.path.join(ROOT, fname)).read().splitlines()}
        for line in body.splitlines():
            n = _norm(line)
            if not n or n.startswith("/* ...") or n == "...":
                continue
            assert n in src, "INTEGRATION.md quotes a line that is not in %s: %r" % (fname, line)
    # the member the round-2 document invented must not come back
    assert "fp_dipole" not in md
