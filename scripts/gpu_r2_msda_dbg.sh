#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for dbg in ${DBGS:-0 1 2}; do
  for wgs in ${WGS:-16}; do
  RSCOTR_MSDA_TILE_RUN=$wgs RSCOTR_MSDA_TILE_DBG=$dbg RSCOTR_MSDA_BWD=tiled timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dbg -o p -- python $R/scripts/bench_msda.py --iters 30 > /tmp/log_$dbg.log 2>&1
  f=$(find /tmp/prof_$dbg -name '*kernel_stats.csv' | head -1)
  echo "== dbg $dbg wgs $wgs"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda_tile' in r['Name']:
        print(f"    {float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']}  {r['Name'][:70]}")
PY
  done
done
