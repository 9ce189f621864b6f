#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for t in cls det seg; do timeout 300 python scripts/bwd_census.py $t > gpurun_out/r2_bwd_census_$t.txt 2>&1; done
tail -5 gpurun_out/r2_bwd_census_seg.txt
