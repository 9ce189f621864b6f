python -m pytest tests/test_mha_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
bash scripts/gpu_ab_bench.sh ab_attn1 "RSCOTR_ATTN_CORE=0" "" "RSCOTR_ATTN_CORE=0" ""
