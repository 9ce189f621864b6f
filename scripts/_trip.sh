mkdir -p gpurun_out/t2
python -m pytest tests/test_h3_gpu.py -q -m gpu -p no:cacheprovider -k "optimizer_keeps" 2>&1 | tail -3 > gpurun_out/t2/opt.log
python -m pytest tests/test_norm_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/t2/pm.log
python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/t2/dist.log
python scripts/lab/dist_overlap_flake.py 24 0 > gpurun_out/t2/flake.log 2>&1
tail -3 gpurun_out/t2/opt.log; tail -3 gpurun_out/t2/pm.log; tail -5 gpurun_out/t2/dist.log; grep -c ": ok" gpurun_out/t2/flake.log; tail -1 gpurun_out/t2/flake.log
