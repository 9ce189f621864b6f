mkdir -p gpurun_out/r4d2
python -m pytest "tests/test_model_gpu.py::test_mlvl_cls_head_variant" "tests/test_sizes_gpu.py::test_swin_b_1024_step_matches_oracle" tests/test_gemm_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/r4d2/test.log; tail -5 gpurun_out/r4d2/test.log
ROWS=160 RSCOTR_PROF_SHAPES=1 python scripts/gemm_shapes.py > gpurun_out/r4d2/gemm_shapes.txt 2>gpurun_out/r4d2/gemm_shapes.err
head -3 gpurun_out/r4d2/gemm_shapes.txt
python bench.py --no-cpu-baseline > gpurun_out/r4d2/bench.json 2>gpurun_out/r4d2/bench.err; cut -c1-1500 gpurun_out/r4d2/bench.json
