mkdir -p gpurun_out/t25
bash scripts/gpu_prof_graph.sh t25/t25 > /dev/null 2>&1
