#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the few-row FFN launches, hidden width cut into runs against one run and the two-product route:
#   bash scripts/lab/ffn_split_lab.sh  -> gpurun_out/ffn_split_lab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/ffn_split_lab.txt; : > $out
for shape in "2048 384 1536 gelu" "1600 256 2048 relu" "2048 192 768 gelu"; do
  for sp in 0 1; do
    rm -rf /tmp/fsl
    RSCOTR_FFN_SPLITS=$sp RSCOTR_FFN_FUSED_MIN_ROWS=1024 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fsl -o t -- python $R/scripts/lab/ffn_cold.py $shape > /tmp/fsl.log 2>&1
    echo "== $shape RSCOTR_FFN_SPLITS=$sp" >> $out
    grep -h '^{' /tmp/fsl.log >> $out
    f=$(find /tmp/fsl -name '*kernel_stats.csv' | head -1)
    python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('ffn_h3', 'gemm_h3', 'gemm_f32', 'splitk_reduce', 'gemm_small')):
        print(f"   {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:7.1f} us  {n[:110]}")
PY
  done
done
cat $out
