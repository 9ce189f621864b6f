#!/bin/bash
# L2 / fabric counters of the grouped weight-gradient launch inside the step (where its 2 GB are served from):
#   bash scripts/lab/pmc_group.sh  -> gpurun_out/r5_group_pmc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/r5_group_pmc.txt; mkdir -p $R/gpurun_out; : > $out
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE WRITE_SIZE"; do
  rm -rf /tmp/pmcg
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "gemm_h3_group_kernel|splitk_flush_kernel" --output-format csv -d /tmp/pmcg -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  f=$(find /tmp/pmcg -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" >> $out <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'].replace('void rscotr::', '').replace('rscotr::', '')[:40], r['Grid_Size'], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, g, c), (n, s) in agg.items():
    print(f'{k:40s} grid {g:>9s} {c:28s} {s / n:16.0f} per launch ({n})')
PY
done
cat $out
