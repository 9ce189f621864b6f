python bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; cut -c1-220 gpurun_out/r4g_bench.json
bash scripts/gpu_prof_graph.sh r4g
python -m pytest tests/test_msda_gpu.py tests/test_model_gpu.py tests/test_sizes_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | cut -c1-200
