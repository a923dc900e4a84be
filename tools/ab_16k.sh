#!/bin/bash
# Production flags at 16 384 atoms on one box: this tree with 3 / 2 / 4 lags and round 2's chain kernel (tools/ab/libmpmc_hip_tnb.so)
lib=mpmc_amd/csrc/libmpmc_hip.so
cp $lib /tmp/ab_tree.so
run() { env $2 python bench.py --workload spolprod_16384 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-22s %9.1f steps/s  chain %.1f us' % (sys.argv[1], d['value'], 1e3 * r.get('avg_launch_ms', 0)))
" "$1"; }
run "tree lags3" "X=1"
run "tree lags2" "MPMC_GS_LAGS=2"
cp tools/ab/libmpmc_hip_tnb.so $lib; run "r02chain" "X=1"; cp /tmp/ab_tree.so $lib
run "tree lags4" "MPMC_GS_LAGS=4"
