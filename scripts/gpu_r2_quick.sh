#!/bin/bash
# quick trip: the listed tests (default: glue + model parity + determinism), then the graph-replay bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-r2q}
TESTS=${2:-"tests/test_glue_gpu.py tests/test_model_gpu.py tests/test_determinism_gpu.py"}
( time timeout 1500 python -m pytest $TESTS -x -q ) > $O/${TAG}_tests.log 2>&1
tail -6 $O/${TAG}_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python - <<PY
import json
ok = False
for line in open('$O/${TAG}_bench.json'):
    if line.startswith('{'):
        d = json.loads(line); ok = True
        print('BENCH', round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'])
if not ok:
    print(open('$O/${TAG}_bench.err').read()[-2500:])
PY
