"""Register budget of the built kernels, read from the code object inside libmpmc_hip.so (no GPU needed).

Two kernels must never spill or use scratch:
  * gs_block_inverse_kernel fetches its tensors with inline-asm LDS loads the compiler cannot see (kernels_gs_chain.h,
    lds_tensor_request / _wait); a register the compiler spills between the request and the wait is stored before its data
    have arrived.  A 16-wave build with five interleaved chains did exactly that (7-11 spilled registers): results differed
    in the last digits from run to run, which only the interleaved-walkers GPU test noticed, two runs in four.
  * gs_chain_kernel keeps P_t (144 registers) resident next to a source loop sized to the last register: a spill there is a
    scratch round trip inside the sweep's critical section (round 3 saw 72 spilled registers cost 8 us per sweep).
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mpmc_amd", "csrc", "libmpmc_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernel_notes(tmp_path_factory):
    d = tmp_path_factory.mktemp("codeobj")
    fat, co = str(d / "fat.bin"), str(d / "mpmc.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--input=" + fat, "--output=" + co, "--unbundle"])
    text = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    kernels = {}
    for block in text.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        field = lambda f: int(re.search(r"\.%s:\s+(\d+)" % f, block).group(1))
        kernels[name.group(1)] = dict(scratch=field("private_segment_fixed_size"), vgpr_spills=field("vgpr_spill_count"),
                                      sgpr_spills=field("sgpr_spill_count"), vgprs=field("vgpr_count"))
    return kernels


def test_code_object_lists_the_kernels(kernel_notes):
    assert len(kernel_notes) > 40
    assert sum("gs_block_inverse_kernel" in k for k in kernel_notes) == 6  # ORTHO x {4, 8, 16} waves
    assert sum("gs_chain_kernel" in k for k in kernel_notes) == 2


@pytest.mark.parametrize("kernel", ["gs_block_inverse_kernel", "gs_chain_kernel"])
def test_no_scratch_and_no_spilled_vector_registers(kernel_notes, kernel):
    hits = {k: v for k, v in kernel_notes.items() if kernel in k}
    assert hits
    for name, r in hits.items():
        assert r["scratch"] == 0 and r["vgpr_spills"] == 0, (name, r)
