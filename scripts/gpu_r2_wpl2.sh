#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
RSCOTR_WPLANES=1 timeout 600 python scripts/gemm_shapes.py > $O/r2_census_wpl1.txt 2>&1
RSCOTR_WPLANES=0 timeout 600 python scripts/gemm_shapes.py > $O/r2_census_wpl0.txt 2>&1
head -3 $O/r2_census_wpl1.txt $O/r2_census_wpl0.txt
