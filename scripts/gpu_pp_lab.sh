#!/bin/bash
# planes x planes GEMM lab on the GPU box: scripts/lab/pp_lab (built here with hipcc, travels with the snapshot).
# usage: gpu_pp_lab.sh <tag> [probe|check|time|all]
tag=${1:-pp_lab}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
timeout 600 ./scripts/lab/pp_lab ${2:-all} ${3:-} > $out/pp_lab.txt 2>&1
echo "rc=$?" >> $out/pp_lab.txt
tail -120 $out/pp_lab.txt
