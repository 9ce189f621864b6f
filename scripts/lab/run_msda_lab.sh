cd scripts/lab && ./msda_lab 0.05 && ./msda_lab -1
cd ../.. && python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py tests/test_determinism_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -4
