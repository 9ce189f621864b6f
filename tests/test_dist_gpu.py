"""The distributed code path on ONE GPU (RSCOTR_DIST_SINGLE=1: a one-rank RCCL group, where every all-reduce(AVG) is the
identity): the captured iterations with the gradient exchange inside the hipGraph — inline on the compute stream (the
default) and forked onto RCCL's stream (RSCOTR_DIST_INLINE=0) — and the split form (RSCOTR_DIST_CAPTURE=0) must train like
the plain run: same losses after three rounds (eager, capture + first replay, replay) from the same seed, up to the rounding of
differently grouped weight-gradient launches (the bucket-aligned flushes cut the grouped launch at bucket boundaries)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
if os.environ.get('RSCOTR_DIST_SINGLE') == '1':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[2])
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('NCCL_DEBUG', 'WARN')
    dist.init_process_group('nccl', device_id=dev)
from rscotr_amd import Config, MODELS
from rscotr_amd.data import build_synthetic_multidataloader
from rscotr_amd.runner import build_runner
cfg = Config.fromfile(os.path.join(sys.argv[1], 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py'))
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(cfg.model); model.init_weights(); model.to(dev).train()
loader = build_synthetic_multidataloader(cfg, dev, size=256, batch_size=2, rank=0)
runner = build_runner(model, cfg, loader)
last = {}
with runner.on_stream():
    for it in range(9):  # three rounds: eager, capture + first replay, replay
        out = runner.train_iter()
        last.update({k: float(v) for k, v in out['log_vars'].items() if k.endswith('.loss')})  # (the task's latest loss)
torch.cuda.synchronize()
norm = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in model.parameters())))
print('RESULT ' + json.dumps(dict(losses=last, param_norm=norm, graphed=sorted(runner.graphed))))
if dist.is_initialized():
    dist.destroy_process_group()
'''


def _run(env_extra, port):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, '-c', _CHILD, ROOT, str(port)], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
    assert lines, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1][7:])


def test_one_rank_distributed_paths_train_like_the_plain_run(cuda):
    plain = _run({'RSCOTR_DIST_SINGLE': '0'}, 29541)
    assert plain['graphed'] == ['cls', 'det', 'seg'] and len(plain['losses']) == 3
    for i, extra in enumerate(({'RSCOTR_DIST_SINGLE': '1'}, {'RSCOTR_DIST_SINGLE': '1', 'RSCOTR_DIST_INLINE': '0'},
                               {'RSCOTR_DIST_SINGLE': '1', 'RSCOTR_DIST_CAPTURE': '0'})):
        got = _run(extra, 29542 + i)
        assert got['graphed'] == plain['graphed'], (extra, got)
        assert abs(got['param_norm'] - plain['param_norm']) <= 1e-6 * plain['param_norm'], (extra, got, plain)
        for k, v in plain['losses'].items():
            assert abs(got['losses'][k] - v) <= 2e-3 * max(abs(v), 1e-3), (extra, k, got['losses'][k], v)
