#!/bin/bash
# idle gaps > 10 us between consecutive kernels in six replayed rounds of the plain run, with the kernels on either side
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_gp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gp -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_gp.log 2>&1
f=$(find /tmp/prof_gp -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
ad = [e for e in ev if 'adamw_clip_kernel' in e[2]]
cut = ad[-19][1]
ev = [e for e in ev if e[0] >= cut]
agg = collections.Counter(); cnt = collections.Counter(); tot = 0
busy_end, prev = ev[0][1], ev[0][2]
for s, e, n in ev[1:]:
    if s > busy_end:
        g = s - busy_end
        tot += g
        if g > 10000:
            k = (prev[:60], n[:60]); agg[k] += g; cnt[k] += 1
    if e >= busy_end:
        busy_end, prev = e, n
span = ev[-1][1] - ev[0][0]
print(f'6 rounds: {span/6e6:.2f} ms per round, idle {tot/6e6:.3f} ms per round; gaps > 10 us by (kernel before -> kernel after), ms per round, count per round:')
for k, v in agg.most_common(25):
    print(f'  {v/6e6:6.3f} ms {cnt[k]/6:5.1f}  {k[0]}  ->  {k[1]}')
PY
