RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_bk32.so RSCOTR_X6_BK0=32 python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
bash scripts/gpu_ab_bench.sh ab_bk32 "" "RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_bk32.so RSCOTR_X6_BK0=32" "" "RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_bk32.so RSCOTR_X6_BK0=32"
