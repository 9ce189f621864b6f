mkdir -p gpurun_out/r3c
python bench.py > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
python bench.py --workload det800 --no-cpu-baseline > gpurun_out/r3c/det800.json 2> gpurun_out/r3c/det800.err
python bench.py --workload swinb1024 --no-cpu-baseline > gpurun_out/r3c/swinb1024.json 2> gpurun_out/r3c/swinb.err
RSCOTR_DIST_SINGLE=1 python bench.py --no-cpu-baseline > gpurun_out/r3c/dist_single.json 2> gpurun_out/r3c/dist.err
for f in bench det800 swinb1024 dist_single; do python - gpurun_out/r3c/$f.json <<'PY'
import json,sys
s=open(sys.argv[1]).read()
i=s.find('{"metric"')
d=json.loads(s[i:].splitlines()[0])
print(sys.argv[1], d['ms_per_step'], d.get('per_task_ms'), d['roofline']['kernel'], round(d['roofline']['frac'],3), 'extra stdout bytes before JSON:', i, 'lines:', len(s.strip().splitlines()))
PY
done
