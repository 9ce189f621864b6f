#!/bin/bash
# MSDA backward A/B under rocprofv3: tiled (TS 16 / 8) vs sorted, per-kernel times at the encoder shape
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/r2_msda_ab.txt
for cfg in "tiled 16" "tiled 8" "sorted 16"; do
  set -- $cfg
  RSCOTR_MSDA_BWD=$1 RSCOTR_MSDA_TS=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -o p -- python $R/scripts/bench_msda.py --iters 30 > /tmp/log_$1_$2.log 2>&1
  echo "== $cfg $(grep fwd_us /tmp/log_$1_$2.log)" >> $R/gpurun_out/r2_msda_ab.txt
  f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -1)
  python - "$f" >> $R/gpurun_out/r2_msda_ab.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda' in r['Name']:
        print(f"    {float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']}  {r['Name'][:70]}")
PY
done
cat $R/gpurun_out/r2_msda_ab.txt
