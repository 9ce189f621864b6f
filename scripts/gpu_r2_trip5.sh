#!/bin/bash
# round 2, trip 5: grouped deferred weight gradients + bf16x6 default: tests, bench, per-kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_optim_gpu.py tests/test_determinism_gpu.py -q > $O/r2t5_units.log 2>&1
tail -8 $O/r2t5_units.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "tiny or 256" > $O/r2t5_model.log 2>&1
tail -5 $O/r2t5_model.log
for g in 1 0; do
  RSCOTR_DW_GROUP=$g timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t5_bench_group$g.json 2> $O/r2t5_bench_group$g.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/r2t5_bench_group$g.json').read().strip().splitlines()[-1])
    print('group$g', round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'], d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'],3)))
except Exception as e:
    print('group$g failed', e); print(open('$O/r2t5_bench_group$g.err').read()[-2500:])
PY
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2t5 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/r2t5_prof.log 2>&1
f=$(find /tmp/prof_r2t5 -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/$O/r2t5_bench_kernel_stats.csv
head -25 "$f" | cut -c1-170
