#!/bin/bash
# SQ / cache counters of ONE GEMM shape (a few launches): bash scripts/lab/pmc_one.sh <tag> M N K a_kmajor b_kmajor
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc_ffn}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$TAG
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_GATE_EN1_sum"; do
  rm -rf /tmp/pmc1
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "ffn_h3" --output-format csv -d /tmp/pmc1 -o p -- python $R/scripts/lab/ffn_cold.py fused > /dev/null 2>&1
  f=$(find /tmp/pmc1 -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" >> $R/gpurun_out/$TAG/counters.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'][:60], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, s) in agg.items():
    print(f'{k:60s} {c:34s} {s / n:16.0f} per launch ({n})')
PY
done
cat $R/gpurun_out/$TAG/counters.txt
