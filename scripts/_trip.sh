python -m pytest tests/test_mha_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head
python scripts/bench_attn.py seg
bash scripts/gpu_ab_bench.sh ab_own "RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_prev.so" "" "RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_prev.so" ""
