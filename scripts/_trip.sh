mkdir -p gpurun_out/t3
python -m pytest tests/test_optim_gpu.py tests/test_seg_loss_gpu.py tests/test_golden_gpu.py tests/test_h3_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['per_task_ms'])"; done
bash scripts/gpu_prof_graph.sh t3/t3 >/dev/null 2>&1; grep -h "adamw_clip\|upsample_ce" gpurun_out/t3/t3_graph_kernel_stats.csv | cut -c1-50,200-330 | head
