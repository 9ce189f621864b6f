set -x
python bench.py > gpurun_out/r6f_bench.json 2> gpurun_out/r6f_bench.err
python bench.py --workload det800 --no-cpu-baseline > gpurun_out/r6f_bench_det800.json 2>> gpurun_out/r6f_bench.err
python bench.py --workload swinb1024 --no-cpu-baseline > gpurun_out/r6f_bench_swinb1024.json 2>> gpurun_out/r6f_bench.err
bash scripts/gpu_prof_graph.sh r6f
bash scripts/gpu_prof_bench.sh r6f
bash scripts/gpu_pmc.sh r6f
ls gpurun_out | grep r6f
