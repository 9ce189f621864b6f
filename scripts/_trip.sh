python -m pytest tests/test_gemm_gpu.py tests/test_h3_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | cut -c1-300
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['per_task_ms'])"; done
