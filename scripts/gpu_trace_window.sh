#!/bin/bash
# consecutive dispatches of one replayed iteration (name, grid, duration, gap to the previous kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-window}
mkdir -p $R/gpurun_out/$T
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_w
env ${2:-} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_w -o t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.log 2>&1
f=$(find /tmp/prof_w -name '*kernel_trace.csv' | head -1)
python - "$f" > $R/gpurun_out/$T/window.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
# last det iteration: find the last lsap_dev_kernel, print 700 kernels before it ... and 900 after
idx = [i for i, r in enumerate(rows) if 'lsap_dev' in r['Kernel_Name']]
c = idx[-1]
lo = max(0, c - 1100)
prev_end = int(rows[lo - 1]['End_Timestamp']) if lo else 0
for r in rows[lo:c + 1200]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    print(f"{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  grid {g:6d}x{r['Grid_Size_Y']:>4s}  lds {r.get('LDS_Block_Size', '?'):>6s}  {r['Kernel_Name'].split('(')[0][:90]}")
    prev_end = e
PY
grep -n "gemm_pp" $R/gpurun_out/$T/window.txt | head -5
