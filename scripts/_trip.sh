python -m pytest tests/test_hplanes_gpu.py tests/test_gemm_gpu.py tests/test_h3_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | cut -c1-300
BENCH_ARGS=--no-roofline bash scripts/gpu_ab_bench.sh t27 "" "RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_no128.so" "" "RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_no128.so" > /dev/null 2>&1
