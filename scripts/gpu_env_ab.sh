#!/bin/bash
# plain bench under several environments: bash scripts/gpu_env_ab.sh "<env A>" "<env B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in "$@"; do
  for rep in 1 2; do
    echo "== $e : $(env $e timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  done
done
