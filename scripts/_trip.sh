python -m pytest tests/test_msda_gpu.py tests/test_golden_gpu.py tests/test_determinism_gpu.py -q -x -m gpu -p no:cacheprovider 2>&1 | grep -E "^E|passed|failed" | head -10
cd scripts/lab; for a in "dec 800 0.5 150" "enc 0.05"; do ./msda_lab $a | grep -E "==|tile kernel  "; done; cd ../..
bash scripts/gpu_ab_bench.sh ab_msda1 "RSCOTR_LIB=$PWD/rscotr_amd/librscotr_old.so" "" "RSCOTR_LIB=$PWD/rscotr_amd/librscotr_old.so" ""
