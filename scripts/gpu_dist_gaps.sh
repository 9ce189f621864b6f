#!/bin/bash
# idle gaps between consecutive kernels (all queues merged) of the one-rank distributed path: where does the wall time go?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for m in 1 0; do
rm -rf /tmp/prof_g$m
RSCOTR_DIST_SINGLE=$m timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g$m -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_g$m.log 2>&1
f=$(find /tmp/prof_g$m -name '*kernel_trace.csv' | head -1)
python - "$f" $m <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows))
# the last 6 rounds only: everything after the 19th-from-last AdamW launch
ad = [e for e in ev if 'adamw_clip_kernel' in e[2]]
cut = ad[-19][1]
ev = [e for e in ev if e[0] >= cut]
busy_end = ev[0][1]
gaps = []
for s, e, n, q, st in ev[1:]:
    if s > busy_end:
        gaps.append((s - busy_end, n, q))
    busy_end = max(busy_end, e)
tot = sum(g[0] for g in gaps)
span = ev[-1][1] - ev[0][0]
print(f'DIST_SINGLE={sys.argv[2]}: 6 rounds, span {span/1e6:.1f} ms = {span/6e6:.2f} ms per round, kernels on queues {collections.Counter(e[3] for e in ev).most_common()},')
print(f'   idle (no kernel on any queue) {tot/1e6:.2f} ms = {100*tot/span:.1f} %, queues {sorted(set(e[3] for e in ev))}')
big = sorted(gaps, reverse=True)[:400]
agg = collections.Counter(); cnt = collections.Counter()
for g, n, q in gaps:
    if g > 3000:
        agg[(n[:70], q)] += g; cnt[(n[:70], q)] += 1
print('gaps > 3 us before kernel (sum ms, count):')
for k, v in agg.most_common(14):
    print(f'  {v/1e6:7.2f} ms {cnt[k]:5d}  q{k[1]}  {k[0]}')
PY
done
