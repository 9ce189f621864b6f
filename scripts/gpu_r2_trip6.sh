#!/bin/bash
# round 2, trip 6: the distributed code path on one rank (RCCL collectives captured in the hipGraph) vs plain
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2t6_bench_plain.json 2> $O/r2t6_bench_plain.err
RSCOTR_DIST_SINGLE=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2t6_bench_dist_capture.json 2> $O/r2t6_bench_dist_capture.err
RSCOTR_DIST_SINGLE=1 RSCOTR_DIST_CAPTURE=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2t6_bench_dist_split.json 2> $O/r2t6_bench_dist_split.err
for w in plain dist_capture dist_split; do python - <<PY
import json
try:
    d = json.loads(open('$O/r2t6_bench_$w.json').read().strip().splitlines()[-1])
    print('$w', round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'], d['config'].get('rccl_ranks'))
except Exception as e:
    print('$w failed', e)
print(open('$O/r2t6_bench_$w.err').read()[-1200:])
PY
done
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_optim_gpu.py -q 2>&1 | tail -4
