mkdir -p gpurun_out/t24
RSCOTR_BF16X6_KMIN=32 python scripts/lab/h3_ksweep.py > gpurun_out/t24/ksweep.txt 2>&1
