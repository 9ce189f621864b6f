RSCOTR_DIST_SINGLE=1 python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r3_ds.json 2> gpurun_out/r3_ds.err; tail -c 1500 gpurun_out/r3_ds.err; head -c 300 gpurun_out/r3_ds.json
bash scripts/gpu_ab_bench.sh r3_grp "" "RSCOTR_DW_GROUP_X6=1" "RSCOTR_DW_GROUP_X6=3" "RSCOTR_DW_GROUP_MAX=70000" "RSCOTR_DW_GROUP_WGS=6144"
