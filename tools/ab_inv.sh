#!/bin/bash
# timing-only builds of the block-inverse kernel (tools/ab/libmpmc_hip_inv{1,2}.so) against the tree's, by in-kernel stamps
lib=mpmc_amd/csrc/libmpmc_hip.so
cp $lib /tmp/ab_tree.so
echo "== tree"; python tools/inv_stamps.py 4096 2>&1 | grep INV_STAMPS | tail -2
for v in 1 2; do cp tools/ab/libmpmc_hip_inv$v.so $lib; echo "== ablate $v"; python tools/inv_stamps.py 4096 2>&1 | grep "INV_STAMPS\|rror" | tail -2; done
cp /tmp/ab_tree.so $lib
