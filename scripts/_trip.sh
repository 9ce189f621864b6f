mkdir -p gpurun_out/t3
RSCOTR_GEMM_H3=1 python -m pytest tests/test_sizes_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "matches_oracle or main_config or mlvl" 2>&1 | tail -40 > gpurun_out/t3/sizes_h3.log
python -m pytest tests/test_sizes_gpu.py -q -m gpu -p no:cacheprovider -k "swin_b_1024" 2>&1 | tail -25 > gpurun_out/t3/sizes_default.log
grep "tests/test_\|passed\|failed" gpurun_out/t3/sizes_h3.log | cut -c1-420
echo ==== default
grep "tests/test_\|passed\|failed" gpurun_out/t3/sizes_default.log | cut -c1-420
