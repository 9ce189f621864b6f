#!/bin/bash
# round 2, trip 1: validate the deterministic reductions, measure the parity statistics against the fp64 oracle, run
# BASELINE configs[3] / configs[4] at size, first bench lines of the three workloads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/r2t1_host.txt; free -g >> $O/r2t1_host.txt
timeout 900 python -m pytest tests/test_neck_gpu.py tests/test_seg_loss_gpu.py tests/test_swin_attn_gpu.py tests/test_optim_gpu.py tests/test_golden_gpu.py -x -q > $O/r2t1_units.log 2>&1
tail -3 $O/r2t1_units.log
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "tiny or 256" > $O/r2t1_model.log 2>&1
tail -3 $O/r2t1_model.log
timeout 1200 python -m pytest tests/test_determinism_gpu.py -q > $O/r2t1_determinism.log 2>&1
tail -15 $O/r2t1_determinism.log
timeout 1500 python scripts/parity_stats.py --sizes 256,512 > $O/r2t1_parity_stats_p0.jsonl 2> $O/r2t1_parity_stats_p0.err
timeout 900 python scripts/parity_stats.py --sizes 512 --prec 2 > $O/r2t1_parity_stats_p2.jsonl 2> $O/r2t1_parity_stats_p2.err
cat $O/r2t1_parity_stats_p0.jsonl $O/r2t1_parity_stats_p2.jsonl | cut -c1-600
timeout 3000 python -m pytest tests/test_sizes_gpu.py -q > $O/r2t1_sizes.log 2>&1
tail -25 $O/r2t1_sizes.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t1_bench_mtl512.json 2> $O/r2t1_bench_mtl512.err
timeout 600 python bench.py --workload det800 --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t1_bench_det800.json 2> $O/r2t1_bench_det800.err
timeout 600 python bench.py --workload swinb1024 --steps 10 --warmup 3 --no-cpu-baseline > $O/r2t1_bench_swinb1024.json 2> $O/r2t1_bench_swinb1024.err
for w in mtl512 det800 swinb1024; do python - <<PY
import json
try:
    d = json.loads(open('$O/r2t1_bench_$w.json').read().strip().splitlines()[-1])
    print('$w', d['value'], d['ms_per_step'], d['per_task_ms'], d['roofline'] and (d['roofline']['kernel'], d['roofline']['frac']))
except Exception as e:
    print('$w failed', e); print(open('$O/r2t1_bench_$w.err').read()[-1500:])
PY
done
