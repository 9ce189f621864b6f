#!/bin/bash
# full GPU suite + default bench (with CPU baseline) — what the driver runs at round end
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
TAG=${1:-r2full}
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/${TAG}_tests.log 2>&1
tail -15 $O/${TAG}_tests.log
( time timeout 900 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${TAG}_smoke.log 2>&1
tail -5 $O/${TAG}_smoke.log
( time timeout 900 python bench.py ) > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -3 $O/${TAG}_bench.err
python - <<PY
import json
for line in open('$O/${TAG}_bench.json'):
    if line.startswith('{'):
        d = json.loads(line)
        print(round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'], d['dtype'][:20])
        print(d['roofline'] and {k: d['roofline'][k] for k in ('kernel','achieved','peak','frac','traffic')})
        print('family', d['roofline_gemm_family'] and {k: d['roofline_gemm_family'][k] for k in ('achieved','peak','frac')})
        print('msda', d['roofline_msda_fwd'] and round(d['roofline_msda_fwd']['frac'],3), d['roofline_msda_bwd'] and (round(d['roofline_msda_bwd']['frac'],3), round(d['roofline_msda_bwd']['avg_us'],1)))
        print('cpu', d['cpu_baseline'])
PY
