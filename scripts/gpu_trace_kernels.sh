#!/bin/bash
# per-dispatch kernel trace of replayed graphs, summarised per (kernel, grid) for the kernels whose name matches $2:
#   scripts/gpu_trace_kernels.sh <tag> <regex> ["ENV=.. ENV=.."]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-trace}; RX=${2:-msda}
mkdir -p $R/gpurun_out/$T
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_t
env ${3:-} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/$T/bench.log 2>&1
f=$(find /tmp/prof_t -name '*kernel_trace.csv' | head -1)
python - "$f" "$RX" > $R/gpurun_out/$T/by_grid.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
rx = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    if not rx.search(name):
        continue
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1) * max(int(r['Grid_Size_Y']), 1)
    k = (re.sub(r'^void ', '', name).split('(')[0][:80], g)
    agg[k][0] += 1; agg[k][1] += dur
for (name, g), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{us:10.0f} us {c:6d} calls {us / c:8.1f} us/call  grid {g:6d}  {name}')
PY
cat $R/gpurun_out/$T/by_grid.txt
