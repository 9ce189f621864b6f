bash scripts/gpu_suite.sh r4s3 > /dev/null 2>&1
tail -2 gpurun_out/r4s3/pytest.log; cat gpurun_out/r4s3/bench.json | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
