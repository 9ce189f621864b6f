#!/bin/bash
# MSDA backward A/B under rocprofv3: tile accumulation vs sorted, per-kernel times at the encoder shape
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/r2_msda_ab.txt
for cfg in ${CFGS:-tiled sorted}; do
  RSCOTR_MSDA_BWD=$cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python $R/scripts/bench_msda.py --iters 30 > /tmp/log_$cfg.log 2>&1
  echo "== $cfg $(grep fwd_us /tmp/log_$cfg.log)" >> $R/gpurun_out/r2_msda_ab.txt
  f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1)
  python - "$f" >> $R/gpurun_out/r2_msda_ab.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda' in r['Name']:
        print(f"    {float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']}  {r['Name'][:70]}")
PY
done
cat $R/gpurun_out/r2_msda_ab.txt
