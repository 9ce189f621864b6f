#!/bin/bash
# MSDA backward A/B on the GPU box: parity of the kernels (tests), then per-kernel times of the micro-benchmark
# (rocprofv3 --kernel-trace --stats) for the tile variants.  Logs under gpurun_out/$1.
tag=${1:-msda_lab}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -15 > $out/pytest.log
tail -3 $out/pytest.log
cd /tmp; export TMPDIR=/tmp
for v in 1 2; do
  RSCOTR_MSDA_TILE_VARIANT=$v python $R/scripts/bench_msda.py --iters 50 > $out/bench_v$v.json 2>&1
  cat $out/bench_v$v.json | tail -1
  rm -rf /tmp/prof_v$v
  RSCOTR_MSDA_TILE_VARIANT=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_v$v -o p --output-format csv -- python $R/scripts/bench_msda.py --iters 20 > /dev/null 2>&1
  f=$(find /tmp/prof_v$v -name '*kernel_stats.csv' | head -1)
  grep -i "msda" "$f" | cut -c1-200 > $out/kernels_v$v.csv
  cat $out/kernels_v$v.csv
done
