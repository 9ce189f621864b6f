#!/bin/bash
# round 2, trip 2: bf16x6 lab (speed + error of the 3-plane split product), parity statistics with the ReLU flip band
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 600 scripts/lab/bf16x6_lab > $O/r2t2_bf16x6_lab.txt 2>&1
cat $O/r2t2_bf16x6_lab.txt
timeout 600 python -m pytest tests/test_optim_gpu.py tests/test_sizes_gpu.py -q -k "graphed or graph_replay" > $O/r2t2_tests.log 2>&1
tail -12 $O/r2t2_tests.log
timeout 1500 python scripts/parity_stats.py --sizes 256,512 --seeds 17,11 > $O/r2t2_parity_stats.jsonl 2> $O/r2t2_parity_stats.err
python - <<'PY'
import json
for line in open('gpurun_out/r2t2_parity_stats.jsonl'):
    d = json.loads(line)
    print(d['task'], d['size'], d['seed'], 'over_tight', d['over_tight'], 'ep_med %.1e eo_med %.1e amb_med %.1e' % (d['ep_med'], d['eo_med'], d['amb_med']),
          'ratio', ['%.1f' % x for x in d['ratio_q']], 'band', ['%.2f' % x for x in d['ratio_band_q']], d['seconds'])
    for w in d['worst_band'][:3]:
        print('      ', w[0], 'ep %.2e eo %.2e amb %.2e' % (w[1], w[2], w[3]))
PY
