#!/bin/bash
# kernel statistics of the default bench command itself (timed graph replays + the eager roofline rounds): the averages that
# bench.py's roofline objects are checked against
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-bench}
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${T}_under_rocprof.json 2> $R/gpurun_out/${T}_prof.err
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/${T}_bench_kernel_stats.csv \;
cut -c1-300 $R/gpurun_out/${T}_under_rocprof.json
head -4 $R/gpurun_out/${T}_bench_kernel_stats.csv | cut -c1-200
