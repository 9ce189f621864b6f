R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/ffn_split_sweep2.txt; : > $out
run() {
  rm -rf /tmp/fsl
  RSCOTR_FFN_SPLITS=$4 RSCOTR_FFN_FUSED_MIN_ROWS=256 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fsl -o t -- python $R/scripts/lab/ffn_cold.py $1 $2 $3 fused flush > /tmp/fsl.log 2>&1
  echo "== $1 x $2 -> $3, runs $4 (cold)" >> $out
  f=$(find /tmp/fsl -name '*kernel_stats.csv' | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('ffn_h3', 'splitk_reduce')):
        print(f"   {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:7.1f} us  {n[:110]}")
PY
}
for sp in 1 2 3; do run 8192 192 768 $sp; done
for sp in 1 3; do run 32768 96 384 $sp; done
for sp in 1 2; do run 10880 256 2048 $sp; done
cat $out
