python -m pytest tests/test_mha_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head
python -m pytest tests/test_model_gpu.py tests/test_determinism_gpu.py tests/test_golden_gpu.py tests/test_inference_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
