#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_profile.sh r01 [extra bench args]
# 1) kernel trace + stats of the default bench command, 2) separate PMC passes for HBM traffic.
set -e
TAG=${1:-rXX}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 50 --warmup 5 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
python3 profiles/summarize.py $OUT $TAG
