#!/bin/bash
# Where the one-rank distributed path's extra time per round goes: per-kernel totals of the replayed graphs, DIST_SINGLE=1 (inline
# / overlap) against the plain run, largest differences first.  -> gpurun_out/$1/dist_diff.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-dist_diff}
mkdir -p $R/gpurun_out/$T
cd /tmp; export TMPDIR=/tmp
for m in plain inline overlap; do
  rm -rf /tmp/prof_d$m
  case $m in plain) env="RSCOTR_DIST_SINGLE=0";; inline) env="RSCOTR_DIST_SINGLE=1";; overlap) env="RSCOTR_DIST_SINGLE=1 RSCOTR_DIST_INLINE=0";; esac
  env $env timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d$m -o p -- python $R/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench_$m.log 2>&1
  find /tmp/prof_d$m -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/$T/${m}_kernel_stats.csv \;
  env $env python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench_${m}.json 2>/dev/null
done
python - $R/gpurun_out/$T <<'PY' > $R/gpurun_out/$T/dist_diff.txt
import csv, json, sys, os
d = sys.argv[1]
def load(m):
    rows = list(csv.DictReader(open(os.path.join(d, f'{m}_kernel_stats.csv'))))
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in rows}
rounds = 36.0  # 2 set-up + 2 warm-up + 30 timed + (capture iterations run eagerly: counted in, same in every arm)
base = load('plain')
for m in ('plain', 'inline', 'overlap'):
    try:
        j = json.load(open(os.path.join(d, f'bench_{m}.json')))
        print(f'{m:8s} {j["ms_per_step"]:.3f} ms per round (unprofiled run), per task {j["per_task_ms"]}')
    except Exception as e:
        print(m, 'bench failed', e)
for m in ('inline', 'overlap'):
    cur = load(m)
    tb, tc = sum(v[1] for v in base.values()), sum(v[1] for v in cur.values())
    print(f'\n== {m}: kernel time {tc / rounds / 1e6:.3f} ms per round against {tb / rounds / 1e6:.3f} plain ({(tc - tb) / rounds / 1e6:+.3f})')
    diff = sorted(((cur.get(k, (0, 0))[1] - base.get(k, (0, 0))[1], k) for k in set(cur) | set(base)), key=lambda x: -abs(x[0]))
    for dv, k in diff[:25]:
        cb, cc = base.get(k, (0, 0)), cur.get(k, (0, 0))
        print(f'  {dv / rounds / 1e3:+9.1f} us/round  calls {cb[0]:6d} -> {cc[0]:6d}  avg {cb[1] / max(cb[0], 1) / 1e3:7.1f} -> {cc[1] / max(cc[0], 1) / 1e3:7.1f} us  {k[:100]}')
PY
cat $R/gpurun_out/$T/dist_diff.txt
