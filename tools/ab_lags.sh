#!/bin/bash
# Same-box comparison of the chain's number of cached lags (option gs_lags): bench lines + the chain kernel with and without its source loop
for rep in 1 2; do
for L in 2 3 4; do
  MPMC_GS_LAGS=$L python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d.get('roofline', {})
        print('gs_lags %s %9.1f steps/s  %s %.2f us' % (sys.argv[1], d['value'], r.get('kernel', '')[:18], 1e3 * r.get('avg_launch_ms', 0)))
" $L
done; done
for L in 2 3 4; do echo "== gs_lags $L"; MPMC_GS_LAGS=$L python tools/gs_ablate.py 4096 0,1 | tail -2; done
