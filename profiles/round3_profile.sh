#!/bin/bash
# Everything profiles/r03_* holds, in one gpurun call (from the repo root on the GPU box):
#   bash profiles/round3_profile.sh > gpurun_out/round3_profile.log 2>&1
# then, back in the build container:  python3 profiles/collect.py r03_jacobi && python3 profiles/collect.py r03_gs gs_chain_kernel
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== headline: stats + PMC"; bash profiles/run_profile.sh r03_jacobi > gpurun_out/prof_r03_jacobi.log 2>&1
echo "== headline bench (with the CPU baseline, 20 s + all cores)"; python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_steps20.json 2>/dev/null
python bench.py --uvt --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r03_bench_uvt.json 2>/dev/null
echo "== every size of the sweep, each with the CPU port timed beside it (8 s, one pinned core)"
: > gpurun_out/r03_bench_other.jsonl
for w in slj_256 ses_1024 spol_1024 spol_4096 spolprod_1024 spolprod_4096; do
  python bench.py --workload $w --steps 1500 --warmup 150 --cpu-budget 8 --cpu-all-cores 0 2>/dev/null | grep "^{" >> gpurun_out/r03_bench_other.jsonl
  echo done $w
done
# 16 384 atoms: the CPU port would need ~46 GB (27 GB pair cache + 19 GB A matrix) and minutes per step: no CPU line
python bench.py --workload spol_16384 --steps 600 --warmup 60 --no-cpu-baseline 2>/dev/null | grep "^{" >> gpurun_out/r03_bench_other.jsonl; echo done spol_16384
# the reference's GPU sample in full (21 183 atoms): NVT and the grand-canonical chain iter.inp asks for
python bench.py --workload pcn61_21183 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | grep "^{" >> gpurun_out/r03_bench_other.jsonl; echo done pcn61_21183
python bench.py --workload pcn61_21183 --uvt --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/r03_bench_pcn61_full_uvt.json; echo done pcn61_21183 uvt
echo "== production flags: stats + PMC + stamps"
bash profiles/run_profile.sh r03_gs --workload spolprod_4096 > gpurun_out/prof_r03_gs.log 2>&1
python bench.py --workload spolprod_4096 --uvt --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/r03_bench_gs_uvt.json 2>/dev/null
python tools/gs_stamps.py > gpurun_out/r03_chain_stamps.txt 2>&1
python tools/inv_stamps.py 4096 2>&1 | grep INV_STAMPS > gpurun_out/r03_inverse_stamps.txt || true
python tools/gs_ablate.py 4096 0,1,2,4 > gpurun_out/r03_chain_ablations.txt 2>&1
bash tools/ab_lib.sh tools/ab/libmpmc_hip_tnb.so r02chain -- --workload spolprod_4096 --steps 1500 --warmup 150 > gpurun_out/r03_ab_chain_4096.txt 2>&1
bash tools/ab_lib.sh tools/ab/libmpmc_hip_tnb.so r02chain -- --workload spolprod_1024 --steps 1500 --warmup 150 > gpurun_out/r03_ab_chain_1024.txt 2>&1
echo "== time line of a production step (kernel trace; host API trace separately)"
rm -rf gpurun_out/tl; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --workload spolprod_4096 --steps 300 --warmup 50 --no-cpu-baseline > /dev/null 2>&1 && python tools/step_timeline.py gpurun_out/tl 200 > gpurun_out/r03_step_timeline.txt; rm -rf gpurun_out/tl
echo "== counters under the VALU kernels"
bash profiles/valu_counters.sh r03_jacobi > gpurun_out/prof_r03_valu.log 2>&1 || echo "valu counters failed"
echo ALL DONE
