python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py tests/test_determinism_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
bash scripts/gpu_ab_bench.sh ab_u3 ""
bash scripts/gpu_pmc.sh r4f > /dev/null 2>&1; ls gpurun_out/r4f_pmc_*csv
