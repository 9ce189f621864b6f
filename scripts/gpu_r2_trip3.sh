#!/bin/bash
# round 2, trip 3: validate bf16x6 (mode 3) and the tiled MSDA backward; determinism; A/B benches
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_msda_gpu.py -q -x > $O/r2t3_units.log 2>&1
tail -8 $O/r2t3_units.log
timeout 900 python -m pytest tests/test_determinism_gpu.py -q > $O/r2t3_determinism.log 2>&1
tail -12 $O/r2t3_determinism.log
timeout 1500 python -m pytest tests/test_model_gpu.py -q > $O/r2t3_model.log 2>&1
tail -12 $O/r2t3_model.log
for mode in fp32 bf16x6; do
  RSCOTR_GEMM_PREC=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t3_bench_$mode.json 2> $O/r2t3_bench_$mode.err
done
RSCOTR_MSDA_BWD=sorted timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t3_bench_sortedmsda.json 2> $O/r2t3_bench_sortedmsda.err
for w in fp32 bf16x6 sortedmsda; do python - <<PY
import json
try:
    d = json.loads(open('$O/r2t3_bench_$w.json').read().strip().splitlines()[-1])
    print('$w', round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'], d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'],3)),
          d['roofline_msda_bwd'] and (round(d['roofline_msda_bwd']['avg_us'],1), round(d['roofline_msda_bwd']['frac'],3)), d['roofline_gemm_family'] and round(d['roofline_gemm_family']['achieved'],1))
except Exception as e:
    print('$w failed', e); print(open('$O/r2t3_bench_$w.err').read()[-1500:])
PY
done
RSCOTR_GEMM_PREC=bf16x6 RSCOTR_PROF_SHAPES=1 timeout 600 python scripts/gemm_shapes.py > $O/r2t3_gemm_census_bf16x6.txt 2>&1
head -70 $O/r2t3_gemm_census_bf16x6.txt
