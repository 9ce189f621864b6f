#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_optim_gpu.py -q 2>&1 | tail -4
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  RSCOTR_DW_GROUP_X6=$1 RSCOTR_BF16X6_MIDSPLIT=$2 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2t9_bench_$1_$2.json 2> $O/r2t9_bench_$1_$2.err
  python - <<PY
import json
try:
    for line in open('$O/r2t9_bench_$1_$2.json'):
        if line.startswith('{'):
            d = json.loads(line); print('x6=$1 midsplit=$2', round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'])
except Exception as e:
    print('failed', e); print(open('$O/r2t9_bench_$1_$2.err').read()[-1500:])
PY
done
