#!/bin/bash
lib=mpmc_amd/csrc/libmpmc_hip.so
cp $lib /tmp/ab_tree.so
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-10s %9.1f steps/s  chain %.2f us' % (sys.argv[1], d['value'], 1e3 * r.get('avg_launch_ms', 0)))
" "$tag"; }
for rep in 1 2 3 4; do
  tag=tree; run "$@"
  cp tools/ab/libmpmc_hip_tnb.so $lib; tag=r02chain; run "$@"; cp /tmp/ab_tree.so $lib
done
