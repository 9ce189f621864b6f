#!/bin/bash
# ordered kernel timeline of ONE eager iteration of a task (the iteration between the last two adamw_clip_kernel launches):
#   bash scripts/gpu_timeline_task.sh <tag> [tasks...]   -> gpurun_out/<tag>_<task>_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-tl}; shift
TASKS=${@:-cls}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for t in $TASKS; do
  rm -rf /tmp/tl_$t
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$t -o $t -- python $R/scripts/profile_task.py $t 4 > $R/gpurun_out/${TAG}_tl_$t.log 2>&1
  f=$(find /tmp/tl_$t -name '*kernel_trace.csv' | head -1)
  python - "$f" > $R/gpurun_out/${TAG}_${t}_timeline.txt <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'adamw_clip_kernel' in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
tot = 0.0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1) * max(int(r['Grid_Size_Y']), 1) * max(int(r['Grid_Size_Z']), 1)
    name = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0][:90]
    tot += (e - s) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  wg {g:6d}  {name}')
print(f'# {b - a} launches, {tot:.1f} us of kernel time, span {(int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3:.1f} us')
PY
done
