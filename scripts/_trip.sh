bash scripts/gpu_suite.sh r4s4 > /dev/null 2>&1
tail -2 gpurun_out/r4s4/pytest.log; cat gpurun_out/r4s4/bench.json | cut -c1-200
bash scripts/gpu_prof_graph.sh r4c
