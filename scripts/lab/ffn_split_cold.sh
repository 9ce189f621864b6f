#!/bin/bash
# few-row FFN launches with everything COLD (768 MB fill between calls, so the weight planes come from HBM as in the step), per library build:
#   bash scripts/lab/ffn_split_cold.sh [lib_variant ...]   ("" = the tree's library; names of rscotr_amd/_ab/lib_<name>.so)  -> gpurun_out/ffn_split_cold.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/ffn_split_cold.txt; : > $out
for v in "" "$@"; do
  IFS=';' read -ra SH <<< "${SHAPES:-2048 384 1536;1600 256 2048;10880 256 2048}"
  for shape in "${SH[@]}"; do
    for fl in flush ""; do
      rm -rf /tmp/fsl
      env ${v:+RSCOTR_LIB=$R/rscotr_amd/_ab/lib_$v.so} RSCOTR_FFN_FUSED_MIN_ROWS=256 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fsl -o t -- python $R/scripts/lab/ffn_cold.py $shape fused $fl > /tmp/fsl.log 2>&1
      echo "== lib '${v:-tree}'  $shape  ${fl:-warm planes}" >> $out
      f=$(find /tmp/fsl -name '*kernel_stats.csv' | head -1)
      python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('ffn_h3', 'splitk_reduce')):
        print(f"   {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:7.1f} us  {n[:110]}")
PY
    done
  done
done
cat $out
