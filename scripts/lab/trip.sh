python -m pytest tests/test_gemm_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -2
echo "== WS=1"; python scripts/bench_gemm.py 2>/dev/null | grep -v "^/opt"
echo "== WS=0"; RSCOTR_BF16X6_WS=0 python scripts/bench_gemm.py 2>/dev/null | grep -v "^/opt"
bash scripts/gpu_ab_bench.sh r3_ws5 "" "RSCOTR_BF16X6_WS=0"
