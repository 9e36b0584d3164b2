#!/bin/bash
# Round-3 evidence, on the GPU box:  bash tools/collect_profiles.sh   (outputs under gpurun_out/; copy what is judged into profiles/)
# rocprofv3 gets the program itself after `--` (python3 bench.py ...), counters in passes of their own with --kernel-trace only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r3prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.json 2> $O/write.err
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_traffic.py $F $W --json $O/pmc_traffic.json > $O/pmc_traffic.txt
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
find $O -name '*.csv' -size +3M -delete; find $O -name '*.db' -delete
ls -la $O
