set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash profiles/run_profile.sh r02_jacobi > gpurun_out/prof_r02_jacobi.log 2>&1
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_steps20.json 2>/dev/null
python bench.py --uvt --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/bench_uvt.json 2>/dev/null
: > gpurun_out/bench_other.jsonl
for w in slj_256 ses_1024 spol_1024 spol_4096 spol_16384 spolprod_1024 spolprod_4096; do
  python bench.py --workload $w --steps 1500 --warmup 150 --no-cpu-baseline 2>/dev/null | grep "^{" >> gpurun_out/bench_other.jsonl
  echo done $w
done
bash profiles/run_profile.sh r02_gs --workload spolprod_4096 > gpurun_out/prof_r02_gs.log 2>&1
python bench.py --workload spolprod_4096 --uvt --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/bench_gs_uvt.json 2>/dev/null
echo ALL DONE
