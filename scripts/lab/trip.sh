python -m pytest tests/test_gemm_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5
bash scripts/gpu_ab_bench.sh r3_edge1 "" "RSCOTR_BF16X6_EDGE=0" "RSCOTR_BF16X6_KMIN=96" "RSCOTR_BF16X6_T64=256" "RSCOTR_BF16X6_KMIN=96 RSCOTR_BF16X6_T64=256"
BENCH_ARGS="--workload det800" bash scripts/gpu_ab_bench.sh r3_edge1_det800 "" "RSCOTR_BF16X6_EDGE=0"
