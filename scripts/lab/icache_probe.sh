#!/bin/bash
# scripts/lab/icache_probe.py under the kernel trace, both modes: gpurun_out/icache_probe.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/icache_probe.txt; mkdir -p $R/gpurun_out; : > $out
for K in 256 1024; do
for m in a b; do
  rm -rf /tmp/icp_$m
  MODE=$m K=$K timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icp_$m -o p -- python $R/scripts/lab/icache_probe.py > /tmp/icp_$m.log 2>&1
  f=$(find /tmp/icp_$m -name '*kernel_stats.csv' | head -1)
  echo "# K=$K MODE=$m" >> $out
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rscotr' in r['Name']:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:7.2f} us min {float(r['MinNs']) / 1e3:7.2f}")
PY
done
done
cat $out
