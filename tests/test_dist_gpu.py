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
    if os.environ.get('RSCOTR_DIST_INLINE') == '0':  # (c10d's flight recorder tells exactly when its watchdog is idle: runner._wait_watchdog_idle)
        os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', sys.argv[2])
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('NCCL_DEBUG', 'WARN')
    dist.init_process_group('nccl', device_id=dev)
from rscotr_amd import Config, MODELS
from rscotr_amd.data import build_synthetic_multidataloader
from rscotr_amd.runner import GraphedTask, build_runner
GraphedTask.capture_veto = os.environ.get('TEST_CAPTURE_VETO') == '1'  # (the test's injection point: not an RSCOTR_* switch of the product)
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
print('RESULT ' + json.dumps(dict(losses=last, param_norm=norm, graphed=sorted(runner.graphed), steps=int(runner.optimizer.steps.sum()),
                                  split=sorted(t for t, g in runner.graphed.items() if g.split))))
if dist.is_initialized():
    dist.destroy_process_group()
'''


def _run(env_extra, port):
    """One child, one try.  (Round 4 retried the overlapped child on c10d's watchdog abort — hipErrorCapturedEvent, 2 of 8 runs.
    The overlapped exchange now runs on a communicator of its own, rscotr_comm_* / dist.DirectComm: no c10d collective is
    captured, the watchdog has nothing of a capturing stream to poll, and a crash is a failure.)"""
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, '-c', _CHILD, ROOT, str(port)], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
    assert lines, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1][7:])


def _loss_tol(key):
    """Relative tolerance of a task's loss after three rounds on two launch groupings of the same step.  The groupings differ in
    the rounding of the weight-gradient sums (1e-7); cls and det carry that to ~1e-3.  The seg decoder thresholds its own mask
    predictions (sigmoid < 0.5 -> attention mask, mask2former_head.py:126-136) in every one of its nine layers: a 1e-7 change of
    the weights flips a few of the ~3 M mask bits of an iteration, and the loss of the THIRD round then differs by 0.1-1 % —
    which side it lands on changes with any re-routing of a GEMM (measured over this round's builds: 1.3e-3, 2.9e-3, 9.4e-3).
    A real fault of these paths (a gradient exchanged twice or not at all, a step count off by one) moves the parameter norm,
    held to 1e-6, and every loss by tens of percent."""
    return 2e-2 if key.startswith('seg.') else 2e-3


def test_one_rank_distributed_paths_train_like_the_plain_run(cuda):
    plain = _run({'RSCOTR_DIST_SINGLE': '0'}, 29541)
    assert plain['graphed'] == ['cls', 'det', 'seg'] and len(plain['losses']) == 3
    for i, extra in enumerate(({'RSCOTR_DIST_SINGLE': '1'}, {'RSCOTR_DIST_SINGLE': '1', 'RSCOTR_DIST_INLINE': '0'},
                               {'RSCOTR_DIST_SINGLE': '1', 'RSCOTR_DIST_CAPTURE': '0'})):
        got = _run(extra, 29542 + i)
        assert got['graphed'] == plain['graphed'], (extra, got)
        assert abs(got['param_norm'] - plain['param_norm']) <= 1e-6 * plain['param_norm'], (extra, got, plain)
        for k, v in plain['losses'].items():
            assert abs(got['losses'][k] - v) <= _loss_tol(k) * max(abs(v), 1e-3), (extra, k, got['losses'][k], v)


def test_capture_fallback_keeps_the_step_counts(cuda):
    """when the ranks agree to drop a captured iteration (one of them failed to capture the RCCL
    collectives), a rank whose capture had SUCCEEDED must start its second attempt from the optimizer state before the first
    one — it has already announced the captured iteration (step counts + 1).  One-rank RCCL group with the veto hook (MIN over
    ranks forced to "failed"): every task ends in the split form, and step counts, weights and losses equal the run that
    took the split form from the start."""
    want = _run({'RSCOTR_DIST_SINGLE': '1', 'RSCOTR_DIST_CAPTURE': '0'}, 29551)
    got = _run({'RSCOTR_DIST_SINGLE': '1', 'TEST_CAPTURE_VETO': '1'}, 29552)
    assert got['graphed'] == want['graphed'] == ['cls', 'det', 'seg']
    assert got['split'] == want['split'] == ['cls', 'det', 'seg']
    assert got['steps'] == want['steps'], (got['steps'], want['steps'])
    # (same trajectory up to the rounding of differently grouped weight-gradient launches, as in the test above; a step count
    # off by one moves the Adam bias corrections of the first steps by tens of percent)
    assert abs(got['param_norm'] - want['param_norm']) <= 1e-6 * want['param_norm'], (got, want)
    for k, v in want['losses'].items():
        if not k.startswith('seg.'):
            continue  # (the second capture attempt consumes more draws of the Mixup / CutMix and denoising-noise streams)
        assert abs(got['losses'][k] - v) <= _loss_tol(k) * max(abs(v), 1e-3), (k, got['losses'][k], v)  # (see _loss_tol)


_CHILD2 = r'''
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
try:
    t = torch.ones(4, device=dev) * (rank + 1)
    dist.all_reduce(t)
    assert float(t[0]) == 3.0
except Exception as e:  # this build's gloo cannot reduce device tensors: nothing to test
    print('RESULT ' + json.dumps(dict(skip=f'gloo all_reduce on device tensors: {type(e).__name__}: {e}')))
    dist.destroy_process_group(); sys.exit(0)
from rscotr_amd import Config, MODELS
from rscotr_amd.data import build_synthetic_multidataloader
from rscotr_amd.runner import build_runner
cfg = Config.fromfile(os.path.join(sys.argv[1], 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py'))
torch.manual_seed(0); np.random.seed(2022)            # identical init and task order on every rank (tools/train.py:211-215)
model = MODELS.build(cfg.model); model.init_weights(); model.to(dev).train()
loader = build_synthetic_multidataloader(cfg, dev, size=256, batch_size=2, rank=rank)   # every rank draws its own batches
runner = build_runner(model, cfg, loader, logger=lambda m: None)
assert runner.sync is not None and runner.ctrl is not None
losses = {}
import warnings
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    with runner.on_stream():
        for it in range(9):   # three rounds: eager (bucket plans), capture in the split form (gloo collectives cannot ride in
            out = runner.train_iter()   # a hipGraph: graph = forward + backward, eager exchange + optimizer) + first replay, replay
            losses.update({k: float(v) for k, v in out['log_vars'].items() if k.endswith('.loss')})
torch.cuda.synchronize()
fell_back = any('falls back' in str(x.message) or 'fall back' in str(x.message) for x in w)
digest = hashlib.sha256(runner.optimizer.flat_p.detach().cpu().numpy().tobytes()).hexdigest()
outs = [None] * world
dist.all_gather_object(outs, dict(digest=digest, losses=losses, graphed=sorted(runner.graphed),
                                  split=sorted(t for t, g in runner.graphed.items() if g.split), fell_back=fell_back))
if rank == 0:
    print('RESULT ' + json.dumps(outs))
dist.destroy_process_group()
'''


@pytest.mark.timeout(1500)
def test_two_ranks_on_one_gpu_train_in_lockstep(cuda):
    """The N > 1 loop end to end with real kernels (the box has one GPU: two gloo ranks share it, each with its own
    batches): bucket plans agreed, gradients averaged, the det capacity and the graph-or-eager decision agreed on through
    the control group, the split form of the captured iteration (graph = forward + backward, then the eager bucket exchange,
    the packed log all-reduce, clip + AdamW), rank-averaged log variables — after three rounds both ranks hold bit-identical
    parameters and report the same (rank-averaged) losses.  RCCL itself needs one GPU per rank: the driver's 8-GPU run."""
    port = 29561
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, '-c', _CHILD2, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, env=env))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=1200))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    lines = [l for l in outs[0][0].splitlines() if l.startswith('RESULT ')]
    assert lines, (outs[0][0][-1500:], outs[0][1][-3000:], outs[1][1][-3000:])
    res = json.loads(lines[-1][7:])
    if isinstance(res, dict) and 'skip' in res:
        pytest.skip(res['skip'])
    a, b = res
    assert a['graphed'] == b['graphed'] == ['cls', 'det', 'seg'] and a['split'] == b['split'] == ['cls', 'det', 'seg']
    assert a['digest'] == b['digest'], 'the ranks hold different parameters after three rounds'
    assert a['losses'] == b['losses'] and len(a['losses']) == 3 and all(v == v and abs(v) < 1e6 for v in a['losses'].values())
