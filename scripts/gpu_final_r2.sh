#!/bin/bash
# Round-2 measurement trip: PMC traffic -> profiles/pmc_gemm_traffic.json, default bench (with CPU baseline), rocprofv3
# kernel statistics of the bench command, the two other BASELINE workloads, the distributed code path on one rank.
# bash scripts/gpu_final_r2.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r2f}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh $T
python scripts/pmc_summary.py gpurun_out/${T}_pmc_FETCH_SIZE.csv gpurun_out/${T}_pmc_WRITE_SIZE.csv profiles/pmc_gemm_traffic.json && cp profiles/pmc_gemm_traffic.json gpurun_out/${T}_pmc_gemm_traffic.json
cd $R
timeout 900 python bench.py --watchdog 800 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -5 gpurun_out/${T}_bench.err | cut -c1-300 > gpurun_out/${T}_bench.err.tail; rm gpurun_out/${T}_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${T}_prof.log 2>&1
find /tmp/prof_b -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/${T}_bench_kernel_stats.csv \;
grep metric $R/gpurun_out/${T}_prof.log | cut -c1-4000 > $R/gpurun_out/${T}_bench_under_rocprof.json; rm $R/gpurun_out/${T}_prof.log
cd $R
for w in det800 swinb1024; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/${T}_bench_$w.json 2> gpurun_out/${T}_bench_$w.err
  tail -3 gpurun_out/${T}_bench_$w.err | cut -c1-300 > gpurun_out/${T}_bench_$w.err.tail; rm gpurun_out/${T}_bench_$w.err
done
RSCOTR_DIST_SINGLE=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_dist_single.json 2> /dev/null
RSCOTR_GEMM_PREC=fp32 timeout 600 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_fp32pipe.json 2> /dev/null
bash scripts/gpu_prof_task.sh $T cls det seg
timeout 120 python scripts/bench_imgprep.py > gpurun_out/${T}_imgprep.json 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob('gpurun_out/${T}_bench*.json')):
    for line in open(f):
        if line.startswith('{') and 'metric' in line:
            d = json.loads(line)
            r = d.get('roofline') or {}
            print(f.split('/')[-1], round(d['value'], 1), round(d['ms_per_step'], 2), d['per_task_ms'], r.get('kernel'), r.get('frac'), r.get('traffic'))
PY
