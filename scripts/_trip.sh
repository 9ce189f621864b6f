mkdir -p gpurun_out/t4
python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -70 > gpurun_out/t4/pytest.log
tail -6 gpurun_out/t4/pytest.log
python bench.py > gpurun_out/t4/bench.json 2> gpurun_out/t4/bench.err
cat gpurun_out/t4/bench.json | cut -c1-3000
