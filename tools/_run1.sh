set -o pipefail
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/gpu_tests.log; cat gpurun_out/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats -o r2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r2_stats.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$grp -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$grp.log 2>&1
done
cd $R
find gpurun_out/r2_stats -name "*kernel_stats.csv" | head -2
cp $(find gpurun_out/r2_stats -name "*kernel_stats.csv" | head -1) gpurun_out/r2_bench_kernel_stats.csv
python tools/pmc_traffic.py $(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) > gpurun_out/r2_pmc_traffic.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/r2_stats
head -20 gpurun_out/r2_bench_kernel_stats.csv; cat gpurun_out/r2_pmc_traffic.txt
