#!/bin/bash
# per-shape GEMM census (scripts/gemm_shapes.py) under two settings: gpu_shapes_ab.sh <tag> "<ENV..>" "<ENV..>"
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$tag
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs python $R/scripts/gemm_shapes.py > $R/gpurun_out/$tag/shapes_$i.txt 2>&1
  echo "== [$envs]"; head -3 $R/gpurun_out/$tag/shapes_$i.txt
done
