#!/bin/bash
# Same-box A/B of two builds of libmpmc_hip.so: tools/ab_lib.sh <other.so> <label> -- <bench.py arguments>
# runs bench.py with the tree's library, then with <other.so> swapped in, then again with the tree's (interleaved pairs
# beat one pair on a noisy box); prints the steps/s and the dominant kernel's mean launch time of each run.
set -e
other=$1; label=$2; shift 3
lib=mpmc_amd/csrc/libmpmc_hip.so
cp $lib /tmp/ab_tree.so
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-10s %9.1f steps/s  %s %.2f us  frac %.3f  fallbacks %s redos %s' % (sys.argv[1], d['value'], r.get('kernel'), 1e3 * r.get('avg_launch_ms', 0), r.get('frac', 0), d.get('resident_fallbacks'), d.get('spec_rank_redos')))
" "$tag"; }
for rep in 1 2; do
  tag=tree; run "$@"
  cp $other $lib; tag=$label; run "$@"; cp /tmp/ab_tree.so $lib
done
