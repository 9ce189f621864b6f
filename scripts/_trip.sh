python -m pytest tests/test_gemm_gpu.py tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
bash scripts/gpu_ab_bench.sh ab_knobs3 "" "RSCOTR_BF16X6_T128=384" "RSCOTR_BF16X6_T128=256" "RSCOTR_BF16X6_KMIN=128" "RSCOTR_BF16X6_KMIN=96" "RSCOTR_DW_GROUP_WGS=3072" "RSCOTR_DW_GROUP_WGS=6144" ""
