#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each) of the fused FFN launch at the encoder shape, per library build:
#   bash scripts/lab/pmc_ffn_traffic.sh [lib_variant ...]   -> gpurun_out/pmc_ffn_traffic.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
out=$R/gpurun_out/pmc_ffn_traffic.txt; : > $out
for v in "" "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc1
    env ${v:+RSCOTR_LIB=$R/rscotr_amd/_ab/lib_$v.so} timeout 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "ffn_h3" --output-format csv -d /tmp/pmc1 -o p -- python $R/scripts/lab/ffn_cold.py 10880 256 2048 fused > /dev/null 2>&1
    f=$(find /tmp/pmc1 -name '*counter_collection.csv' | head -1)
    echo "== lib '${v:-tree}' $c (KiB per launch; HBM read bytes = 2 x FETCH_SIZE KiB on gfx950)" >> $out
    [ -n "$f" ] && python - "$f" >> $out <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'][:60], r['Counter_Name'])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, s) in agg.items():
    print(f'   {k:60s} {c:12s} {s / n:12.0f} ({n})')
PY
  done
done
cat $out
