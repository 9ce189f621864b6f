"""bench.py's launcher: `python bench.py --gpus N` must START N ranks (one process per GPU over RCCL,
what tools/train.py:173-182 / mtl/apis/train.py:37-46 get from `init_dist(args.launcher, ...)`) and print ONE JSON line with
n_gpus == rccl_ranks == N; an inconsistent --gpus / WORLD_SIZE pair must fail loudly instead of running one rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(kw)
    return env


def test_gpus_flag_that_disagrees_with_the_world_size_fails_loudly():
    """(no GPU needed: the check precedes everything else)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--launcher', 'none'],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode != 0 and '--gpus 8' in r.stderr and r.stdout.strip() == ''
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'],
                       capture_output=True, text=True, timeout=300, env=_env(WORLD_SIZE='4', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr and r.stdout.strip() == ''


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_spawn_path_starts_the_ranks_and_prints_one_line(cuda):
    """The N > 1 path of `python bench.py --gpus N` on the one GPU of this box: --launcher spawn re-runs the command under
    torch.distributed.run with one rank, RSCOTR_DIST_SINGLE=1 makes that rank a one-rank RCCL group -> n_gpus == rccl_ranks == 1,
    every task graphed, exactly one JSON line on stdout."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--launcher', 'spawn', '--steps', '2',
                        '--warmup', '1', '--size', '256', '--no-cpu-baseline', '--no-roofline', '--exchange', 'inline'],
                       capture_output=True, text=True, timeout=1100, env=_env(RSCOTR_DIST_SINGLE='1'))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['config']['rccl_ranks'] == 1 and out['config']['exchange'] == 'inline'
    assert out['config']['hipgraph_tasks'] == ['cls', 'det', 'seg'] and out['value'] > 0
    assert 'starting 1 rank(s)' in r.stderr
