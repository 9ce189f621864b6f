python -m pytest tests/test_gemm_gpu.py tests/test_abi_cpu.py -q -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head
bash scripts/gpu_ab_bench.sh ab_tpl3 ""
