# scratch trip script of the session (what `gpurun -- 'bash scripts/_trip.sh'` runs)
mkdir -p gpurun_out
python scripts/lab/h3_lab.py > gpurun_out/h3_lab.txt 2>&1
python scripts/lab/h3_lab.py quick scales tail > gpurun_out/h3_lab_tail.txt 2>&1
tail -30 gpurun_out/h3_lab.txt; tail -10 gpurun_out/h3_lab_tail.txt
