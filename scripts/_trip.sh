bash scripts/gpu_suite.sh r4s6 > /dev/null 2>&1
tail -2 gpurun_out/r4s6/pytest.log | cut -c1-200; cat gpurun_out/r4s6/bench.json | cut -c1-200
bash scripts/gpu_prof_graph.sh r4d
bash scripts/gpu_prof_bench.sh r4d | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
