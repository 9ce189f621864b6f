mkdir -p gpurun_out/t9
RSCOTR_LIB=$PWD/rscotr_amd/_ab/lib_one64.so python scripts/lab/h3_det_flip.py > gpurun_out/t9/flip.txt 2>&1
