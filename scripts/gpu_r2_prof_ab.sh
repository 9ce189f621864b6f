#!/bin/bash
# kernel statistics of the bench command under two environments: bash scripts/gpu_r2_prof_ab.sh "<env A>" "<env B>"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  rm -rf /tmp/prof_ab_$i
  env $e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab_$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_ab_$i.log 2>&1
  f=$(find /tmp/prof_ab_$i -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/r2_prof_ab_$i.csv
  echo "== $e"; grep -o '"ms_per_step": [0-9.]*' /tmp/prof_ab_$i.log
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    print('%8.2f ms %6d x %8.1f us  %s' % (float(r['TotalDurationNs']) / 1e6, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:100]))
PY
done
