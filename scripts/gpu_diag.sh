#!/bin/bash
# one diagnostic trip: a named test with full output, the problems of the grouped weight-gradient launches, the per-shape GEMM
# census of an eager round, and the per-(kernel, grid) durations inside replayed graphs
T=${1:-diag}
out=gpurun_out/$T
mkdir -p $out
if [ -n "$2" ]; then python -m pytest "$2" -x -q -m gpu -p no:cacheprovider 2>&1 | tail -400 > $out/test.log; tail -3 $out/test.log; fi
RSCOTR_DW_GROUP_DUMP=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $out/group_dump.txt 2>$out/group_dump.err
grep -c "M=" $out/group_dump.txt
RSCOTR_PROF_SHAPES=1 python scripts/gemm_shapes.py > $out/gemm_shapes.txt 2>$out/gemm_shapes.err
head -5 $out/gemm_shapes.txt
bash scripts/gpu_trace_pp.sh $T/trace > /dev/null 2>&1
head -30 $out/trace/by_grid.txt
