#!/bin/bash
# SQ / cache counters of the kernels matching a regex in ANY command: bash scripts/lab/pmc_cmd.sh <tag> <kernel regex> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; RX=$2; shift 2
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG; rm -f $R/gpurun_out/$TAG/counters.txt
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc1
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "$RX" --output-format csv -d /tmp/pmc1 -o p -- "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc1 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" >> $R/gpurun_out/$TAG/counters.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'].replace('void rscotr::', '')[:50], r['Grid_Size'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, g, c), (n, s) in agg.items():
    print(f'{k:50s} grid {g:>8s} {c:34s} {s / n:16.0f} per launch ({n})')
PY
done
cat $R/gpurun_out/$TAG/counters.txt
