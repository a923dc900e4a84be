#!/bin/bash
# Same-box comparison of several environment settings, interleaved REPS times:
#   tools/ab_multi.sh REPS "VAR=a" "VAR=b VAR2=c" ... -- <bench.py arguments>      ("-" = the defaults)
reps=$1; shift
settings=()
while [ "$1" != "--" ]; do settings+=("$1"); shift; done
shift
for rep in $(seq $reps); do
  for s in "${settings[@]}"; do
    if [ "$s" = "-" ]; then e=""; else e="$s"; fi
    env $e python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-40s %9.1f steps/s  %s %.2f us' % (sys.argv[1], d['value'], r.get('kernel', '')[:18], 1e3 * r.get('avg_launch_ms', 0)))
" "$s"
  done
done
