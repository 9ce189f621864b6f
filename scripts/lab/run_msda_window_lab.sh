#!/bin/bash
# MSDA forward with LDS value windows against the shipped kernel (scripts/lab/msda_window_lab.hip): parity, LDS hit rates, times
#   bash scripts/lab/run_msda_window_lab.sh   -> gpurun_out/msda_window_lab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/msda_window_lab.txt; : > $out
cd $R && bash scripts/lab/build_msda_window_lab.sh >> $out 2>&1 || { cat $out; exit 1; }
cd /tmp; export TMPDIR=/tmp
for args in "init 0.3" "init 1.0" "spread 0.02" "spread 0.05" "init 0.3 160 96 64 36" "init 0.3 324 196 144 121"; do
  echo "== msda_window_lab $args" >> $out
  $R/scripts/lab/msda_window_lab $args >> $out 2>&1
  rm -rf /tmp/mwl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mwl -o t -- $R/scripts/lab/msda_window_lab $args > /dev/null 2>&1
  f=$(find /tmp/mwl -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'msda' in r['Name']:
        print(f"   rocprofv3: {int(r['Calls']):4d} x {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:100]}")
PY
done
cat $out
