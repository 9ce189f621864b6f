#!/bin/bash
# SQ counters of the MSDA tile kernel (micro-benchmark): instruction mix and wait cycles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  RSCOTR_MSDA_BWD=tiled timeout 300 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex 'msda_tile_kernel' --output-format csv -d /tmp/pmcm_$i -o p -- python $R/scripts/bench_msda.py --iters 5 > /tmp/pmcm_$i.log 2>&1
  f=$(find /tmp/pmcm_$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
try:
    for r in csv.DictReader(open(sys.argv[1])):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
except Exception as e:
    print('no data', e)
for k, v in agg.items():
    print(f'{k:26s} {sum(v)/len(v):16.0f} per launch ({len(v)} launches)')
PY
done
