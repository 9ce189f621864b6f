#!/bin/bash
# round 2, trip 4: deterministic sorted MSDA backward + tuned bf16x6 dispatch: tests, A/B benches, census
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
bash scripts/gpu_r2_msda_ab.sh 2>&1 | grep -A9 "== sorted"
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_msda_gpu.py tests/test_determinism_gpu.py tests/test_optim_gpu.py -q > $O/r2t4_units.log 2>&1
tail -6 $O/r2t4_units.log
for mode in fp32 bf16x6; do
  RSCOTR_GEMM_PREC=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t4_bench_$mode.json 2> $O/r2t4_bench_$mode.err
done
for w in fp32 bf16x6; do python - <<PY
import json
try:
    d = json.loads(open('$O/r2t4_bench_$w.json').read().strip().splitlines()[-1])
    print('$w', round(d['value'],1), round(d['ms_per_step'],2), d['per_task_ms'], d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'],3)),
          d['roofline_msda_bwd'] and (round(d['roofline_msda_bwd']['avg_us'],1), round(d['roofline_msda_bwd']['frac'],3)), d['roofline_gemm_family'] and round(d['roofline_gemm_family']['achieved'],1))
except Exception as e:
    print('$w failed', e); print(open('$O/r2t4_bench_$w.err').read()[-1500:])
PY
done
RSCOTR_GEMM_PREC=bf16x6 RSCOTR_PROF_SHAPES=1 timeout 600 python scripts/gemm_shapes.py > $O/r2t4_gemm_census_bf16x6.txt 2>&1
head -3 $O/r2t4_gemm_census_bf16x6.txt
RSCOTR_DIST_SINGLE=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/r2t4_bench_distsingle.json 2> $O/r2t4_bench_distsingle.err
tail -c 400 $O/r2t4_bench_distsingle.json; tail -5 $O/r2t4_bench_distsingle.err
