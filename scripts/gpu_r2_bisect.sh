#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
T='tests/test_sizes_gpu.py::test_swin_b_1024_step_matches_oracle[seg]'
for e in "RSCOTR_SOFTMAX_VEC=1" "RSCOTR_SOFTMAX_VEC=2" "RSCOTR_SOFTMAX_VEC=$((3 + 16*101))" "RSCOTR_SOFTMAX_VEC=$((3 + 16*257))" "RSCOTR_SOFTMAX_VEC=$((3 + 16*1025))"; do
  echo "== $e: $(env $e timeout 600 python -m pytest "$T" -x -q 2>&1 | grep -E "passed|failed|over_tight" | cut -c1-150 | tail -2 | tr '\n' ' ')"
done
