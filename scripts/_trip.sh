mkdir -p gpurun_out/t26
RSCOTR_HPLANES=1 RSCOTR_HPLANES_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/t26/dbg.txt >/dev/null
grep -c hplanes gpurun_out/t26/dbg.txt
