python -m pytest tests/test_glue_gpu.py tests/test_determinism_gpu.py -q -m gpu -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head
bash scripts/gpu_trace_kernels.sh tr_lvl "level_embed|gn_finalize|focal_sum|gap_tokens|cdn_embed" > /dev/null; cat gpurun_out/tr_lvl/by_grid.txt; grep ms_per gpurun_out/tr_lvl/bench.log | cut -c1-100
