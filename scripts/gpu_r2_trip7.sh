#!/bin/bash
# round 2, trip 7: the rest of the GPU suite (without -x), ATen op attribution per task
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( time timeout 3000 python -m pytest tests/ -q -m gpu ) > $O/r2t7_tests.log 2>&1
tail -15 $O/r2t7_tests.log
for t in det seg cls; do timeout 300 python scripts/op_sources.py $t > $O/r2t7_ops_$t.txt 2>&1; head -45 $O/r2t7_ops_$t.txt; done
