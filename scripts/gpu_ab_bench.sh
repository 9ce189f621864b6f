#!/bin/bash
# A/B of bench.py under environment settings on ONE box: scripts/gpu_ab_bench.sh <tag> "<ENV=.. ENV=..>" "<...>" ...
# (each argument = one variant's environment; "" = defaults).  Prints ms_per_step and per-task ms per variant.
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $out/v$i.json 2> $out/v$i.err
  python - "$out/v$i.json" "$envs" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    fam = d.get('roofline_gemm_family') or {}
    print(f"[{sys.argv[2] or 'defaults'}] {d['ms_per_step']:.2f} ms/step  per-task {d.get('per_task_ms')}  "
          f"split share {fam.get('split_product_flop_share')}  dominant {(d.get('roofline') or {}).get('kernel')} frac {(d.get('roofline') or {}).get('frac')}")
except Exception as e:
    print(f"[{sys.argv[2]}] FAILED {e}"); print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
PY
done
