#!/bin/bash
# HBM traffic counters of the bench command, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do
# not fit one pass; no trace domains besides --kernel-trace next to --pmc).  Counter collection serialises every
# instrumented dispatch (~11 ms each here), so it is restricted to the tiled GEMM family (fp32 pipe, the fp16 / bf16 split products, the grouped weight-gradient launch) and the MSDA kernels: the whole
# step (80k dispatches) does not finish in 15 minutes per counter.  scripts/pmc_summary.py folds the two CSVs into
# profiles/pmc_gemm_traffic.json, which bench.py reads for roofline.traffic.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r1}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 420 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex 'gemm_f32_kernel|gemm_h3|ffn_h3|lin_h3|gemm_bf16x6|gemm_f32_group|msda_|attn_' --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --roofline-rounds 1 > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  python - "$f" "$R/gpurun_out/${TAG}_pmc_$c.csv" <<'PY'
import csv, sys, collections
src, dst = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
with open(src) as fh:
    for r in csv.DictReader(fh):
        k = (r['Kernel_Name'], r['Counter_Name'])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
with open(dst, 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['Kernel_Name', 'Counter_Name', 'Dispatches', 'Sum', 'Mean'])
    for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, s, s / n])
PY
  tail -2 $R/gpurun_out/${TAG}_pmc_$c.log | cut -c1-300 > $R/gpurun_out/${TAG}_pmc_$c.tail; rm $R/gpurun_out/${TAG}_pmc_$c.log
done
