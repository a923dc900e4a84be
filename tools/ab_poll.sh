#!/bin/bash
lib=mpmc_amd/csrc/libmpmc_hip.so
cp $lib /tmp/ab_tree.so
echo "== tree (3 in flight)"; python tools/gs_ablate.py 4096 0,1 | tail -2
for v in 1 2; do cp tools/ab/libmpmc_hip_poll$v.so $lib; echo "== poll variant $v"; python tools/gs_ablate.py 4096 0,1 | tail -2; done
cp /tmp/ab_tree.so $lib
