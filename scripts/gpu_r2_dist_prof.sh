#!/bin/bash
# kernel statistics of the one-rank distributed code path (RSCOTR_DIST_SINGLE=1) next to the plain run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for m in 0 1; do
  rm -rf /tmp/prof_d$m
  RSCOTR_DIST_SINGLE=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d$m -o p -- python $R/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_d$m.log 2>&1
  f=$(find /tmp/prof_d$m -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/r2_dist_prof_$m.csv
  echo "== DIST_SINGLE=$m"; grep -o '"ms_per_step": [0-9.]*' /tmp/prof_d$m.log
done
python - <<PY
import csv
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a = load('$R/gpurun_out/r2_dist_prof_0.csv'); b = load('$R/gpurun_out/r2_dist_prof_1.csv')
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append(((tb - ta) / 1e6, k, ca, cb, ta / 1e6, tb / 1e6))
print('total ms', sum(v[1] for v in a.values()) / 1e6, sum(v[1] for v in b.values()) / 1e6)
for d, k, ca, cb, ta, tb in sorted(rows, key=lambda r: -abs(r[0]))[:25]:
    print('%+8.2f ms  calls %5d -> %5d   %8.2f -> %8.2f ms  %s' % (d, ca, cb, ta, tb, k[:100]))
PY
