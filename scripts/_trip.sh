mkdir -p gpurun_out/t1
python -m pytest tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
for e in "" "RSCOTR_BF16X6_KMIN=96" "" "RSCOTR_BF16X6_KMIN=96"; do env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$e', r['ms_per_step'], r['per_task_ms'])"; done
