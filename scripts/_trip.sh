bash scripts/lab/pmc_cmd.sh pmc_h3_64 'gemm_h3_kernel<64, 64, false, true, 2, false>' python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2>&1
wc -l gpurun_out/pmc_h3_64/counters.txt
