python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
bash scripts/gpu_ab_bench.sh ab_knobs4 "" "RSCOTR_GEMM_SMALL_TILES=1024" "RSCOTR_GEMM_SMALL_K=768" "RSCOTR_GEMM_KG4_MAX=384" "RSCOTR_GEMM_KG2_MAX=1024" "RSCOTR_BF16X6_T128=768" "RSCOTR_GEMM_SPLIT_TARGET=768" "RSCOTR_GEMM_SPLIT_TARGET=384" ""
