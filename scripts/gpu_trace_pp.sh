#!/bin/bash
# per-dispatch kernel trace of replayed graphs; summarised per (kernel, grid size) for the GEMM kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-trace}
mkdir -p $R/gpurun_out/$T
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_t
env ${2:-} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.log 2>&1
f=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
python - "$f" > $R/gpurun_out/$T/by_grid.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
n = 0
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if 'gemm' not in name and 'split_planes' not in name:
        continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    k = (name.split('(')[0][:70], g)
    agg[k][0] += 1; agg[k][1] += dur
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f'total GEMM-family us (whole run): {tot:.0f}')
for (name, g), (c, us) in rows[:70]:
    print(f'{us:10.0f} us {c:6d} calls {us / c:8.1f} us/call  grid {g:6d}  {name}')
PY
head -45 $R/gpurun_out/$T/by_grid.txt
