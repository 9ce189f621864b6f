bash scripts/gpu_suite.sh r4s2 > /dev/null 2>&1
tail -3 gpurun_out/r4s2/pytest.log; cat gpurun_out/r4s2/bench.json | cut -c1-200
bash scripts/gpu_prof_graph.sh r4b
bash scripts/gpu_prof_bench.sh r4b | cut -c1-200
