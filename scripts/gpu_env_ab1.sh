#!/bin/bash
# plain bench under several environments, one run each: bash scripts/gpu_env_ab1.sh "<env A>" "<env B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for e in "$@"; do
  echo "== $e : $(env $e timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|Error.*' | head -2)"
done
