#!/bin/bash
# rocprofv3 kernel stats of one bench workload -> gpurun_out/<tag>_stats.csv (top kernels by total time)
tag=$1; shift
export TMPDIR=/tmp
d=gpurun_out/qs_$tag
rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" > $d/bench.json 2> $d/err.txt
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-90s calls %6s  avg %9.2f us  total %9.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, float(r["Percentage"])))
PY
