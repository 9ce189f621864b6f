mkdir -p gpurun_out/r5
bash scripts/gpu_prof_graph.sh r5/r5
bash scripts/gpu_prof_bench.sh r5/r5
bash scripts/gpu_pmc.sh r5/r5
python scripts/pmc_summary.py gpurun_out/r5/r5_pmc_FETCH_SIZE.csv gpurun_out/r5/r5_pmc_WRITE_SIZE.csv gpurun_out/r5/pmc_gemm_traffic.json
cp gpurun_out/r5/pmc_gemm_traffic.json profiles/pmc_gemm_traffic.json
python bench.py > gpurun_out/r5/r5_bench.json 2> gpurun_out/r5/r5_bench.err; cut -c1-300 gpurun_out/r5/r5_bench.json
python bench.py --workload det800 --no-cpu-baseline > gpurun_out/r5/r5_bench_det800.json 2>/dev/null; cut -c1-300 gpurun_out/r5/r5_bench_det800.json
python bench.py --workload swinb1024 --no-cpu-baseline > gpurun_out/r5/r5_bench_swinb1024.json 2>/dev/null; cut -c1-300 gpurun_out/r5/r5_bench_swinb1024.json
RSCOTR_DIST_SINGLE=1 python bench.py --no-cpu-baseline > gpurun_out/r5/r5_bench_dist_single.json 2>/dev/null; cut -c1-300 gpurun_out/r5/r5_bench_dist_single.json
RSCOTR_DIST_SINGLE=1 python bench.py --no-cpu-baseline --exchange overlap > gpurun_out/r5/r5_bench_dist_single_overlap.json 2>/dev/null; cut -c1-300 gpurun_out/r5/r5_bench_dist_single_overlap.json
