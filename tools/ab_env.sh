#!/bin/bash
# Same-box A/B of an environment setting: tools/ab_env.sh "VAR=value [VAR2=value2]" -- <bench.py arguments>; interleaved pairs
set -e
setting=$1; shift 2
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-34s %9.1f steps/s  %s %.2f us' % (sys.argv[1], d['value'], r.get('kernel', '')[:18], 1e3 * r.get('avg_launch_ms', 0)))
" "$tag"; }
for rep in 1 2; do
  tag=default; run "$@"
  tag="$setting"; env $setting python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('%-34s %9.1f steps/s  %s %.2f us' % (sys.argv[1], d['value'], r.get('kernel', '')[:18], 1e3 * r.get('avg_launch_ms', 0)))
" "$tag"
done
