python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider --tb=short -x -k "presplit" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-300 | head
bash scripts/gpu_ab_bench.sh ab_tpl4 "" "RSCOTR_WPLANES_TILED=1" "" "RSCOTR_WPLANES_TILED=1"
bash scripts/gpu_trace_kernels.sh tr_tpl4 "gemm_bf16x6|split_weights" "RSCOTR_WPLANES_TILED=1" > /dev/null; head -14 gpurun_out/tr_tpl4/by_grid.txt
