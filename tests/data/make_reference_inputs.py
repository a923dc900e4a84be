#!/usr/bin/env python3
"""Copy the reference's own sample INPUT DATA for the two real systems of SURVEY section 8(f3) into tests/data/
(run in the build container, where /root/reference exists; the results are committed):

  sample_configs_gpu/cuda_pol/noncuda_control/socMOF+BSSP.initial.pdb  -> tests/data/socmof/socMOF+BSSP.initial.pdb
      In-soc-MOF + 156 BSSP H2, 1228 atoms + 8 `BOX` marker lines (which the reference's reader skips,
      src/io/read_pqr.c:221-224).  The reference holds this run's output: socMOF+BSSP.energy.dat:2 is the step-0 line.
  sample_configs_gpu/3_PCN61/input.pdb                                   -> tests/data/pcn61_full/input.pdb.gz
      PCN-61 3x1x1 supercell + BSSP H2, 21 183 atoms (6 048 frozen); inputs only, the reference holds no output.

Coordinates and force-field columns are data; nothing of the reference's source is copied.  The keyword files beside
them (tests/data/*/input) are written by hand from the reference's `input` / `iter.inp` with the output paths changed
and `hip on` added; they are not generated here.
"""
import gzip
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/sample_configs_gpu"

shutil.copyfile(os.path.join(REF, "cuda_pol", "noncuda_control", "socMOF+BSSP.initial.pdb"),
                os.path.join(HERE, "socmof", "socMOF+BSSP.initial.pdb"))
with open(os.path.join(REF, "3_PCN61", "input.pdb"), "rb") as f, \
        gzip.GzipFile(os.path.join(HERE, "pcn61_full", "input.pdb.gz"), "wb", compresslevel=9, mtime=0) as g:
    shutil.copyfileobj(f, g)
print("ok")
