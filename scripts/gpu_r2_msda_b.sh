#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for cfg in ${CFGS:-"1 0 16"}; do
  IFS=, read B dbg wgs <<< "$cfg"
  RSCOTR_MSDA_TILE_RUN=$wgs RSCOTR_MSDA_BWD=tiled timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o p -- python $R/scripts/bench_msda.py --iters 20 --B $B > /tmp/log_$cfg.log 2>&1
  f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1)
  python - "$f" "$cfg" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda_tile_kernel' in r['Name']:
        print(f"B,dbg,wgs={sys.argv[2]:10s} {float(r['AverageNs'])/1e3:8.1f} us  tile kernel")
PY
done
