#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for t in cls det seg; do timeout 300 python scripts/aten_census3.py $t > gpurun_out/r2_aten3_$t.txt 2>&1; done
tail -5 gpurun_out/r2_aten3_seg.txt
