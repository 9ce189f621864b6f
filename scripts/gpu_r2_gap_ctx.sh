#!/bin/bash
# the kernels around the largest idle gaps of the replayed rounds (plain run)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_gc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gc -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_gc.log 2>&1
f=$(find /tmp/prof_gc -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
ad = [e for e in ev if 'adamw_clip_kernel' in e[2]]
cut = ad[-10][1]
ev = [e for e in ev if e[0] >= cut]
gaps = []
for i in range(1, len(ev)):
    g = ev[i][0] - max(e[1] for e in ev[max(0, i - 4):i])
    if g > 30000:
        gaps.append((g, i))
for g, i in sorted(gaps, reverse=True)[:8]:
    print(f'--- gap {g/1e3:.0f} us')
    for j in range(max(0, i - 6), min(len(ev), i + 5)):
        mark = '>>' if j == i else '  '
        print(f'  {mark} {(ev[j][1]-ev[j][0])/1e3:7.1f} us  {ev[j][2][:110]}')
PY
