"""Config-driven loop wiring (mtl/apis/train.py:77-118): build_runner honours checkpoint_config, log_config (Text /
Tensorboard logger hooks), evaluation, resume_from / load_from / auto_resume.  Host logic only: a two-head toy model with
the MTL.train_step contract; the optimizer's two HIP launches are replaced by the same arithmetic in torch (the kernels
themselves: tests/test_optim_gpu.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn


class Toy(nn.Module):
    CLASSES = None

    def __init__(self):
        super().__init__()
        self.backbone = nn.Linear(6, 8)
        self.cls_head = nn.Linear(8, 3)
        self.seg_head = nn.Linear(8, 2)

    def train_step(self, data, optimizer=None):
        h = torch.tanh(self.backbone(data['img']))
        y = (self.cls_head if data['task'] == 'cls' else self.seg_head)(h)
        loss = y.pow(2).mean()
        name = f"{data['task']}.{data['dataset_name']}"
        return dict(loss=loss, log_vars={f'{name}.loss': float(loss.detach()), f'{name}.n': float(data['img'].shape[0])},
                    num_samples=data['img'].shape[0])


class Loader:
    """Endless alternating cls / seg batches, deterministic in the iteration index (so a resumed run sees what the
    uninterrupted one saw — the reference's loaders are re-seeded the same way on resume)."""

    def __init__(self, start=0):
        self.i = start

    def __iter__(self):
        return self

    def __next__(self):
        i = self.i
        self.i += 1
        g = torch.Generator().manual_seed(100 + i)
        task = ('cls', 'seg')[i % 2]
        return dict(task=task, dataset_name=dict(cls='resisc', seg='potsdam')[task], img=torch.randn(2 + i % 3, 6, generator=g))


def _torch_step(self, table=None):
    """FlatAdamW.launch_step in torch: clip_grad_norm_(max_norm) + AdamW on the live segments (csrc/optim.hip)."""
    dyn = (self._dyn_host if table is None else table).numpy()
    b1, b2 = self.betas
    g = self.flat_g
    live = torch.zeros(self.total)
    for i, (grp, o) in enumerate(zip(self.groups, self.offsets)):
        if dyn[i, 4]:
            live[o:o + grp['param'].numel()] = 1
    norm = float((g * live).double().pow(2).sum().sqrt())
    scale = min(1.0, self.max_norm / (norm + 1e-6)) if self.max_norm > 0 else 1.0
    with torch.no_grad():
        for i, (grp, o) in enumerate(zip(self.groups, self.offsets)):
            if not dyn[i, 4]:
                continue
            n = grp['param'].numel()
            gi = g[o:o + n] * scale
            p, m, v = self.flat_p[o:o + n], self.flat_m[o:o + n], self.flat_v[o:o + n]
            p.mul_(1 - float(dyn[i, 0]) * float(dyn[i, 1]))
            m.mul_(b1).add_(gi, alpha=1 - b1)
            v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
            p.addcdiv_(m * float(dyn[i, 2]), (v.sqrt() * float(dyn[i, 3])).add_(self.eps), value=-float(dyn[i, 0]))


@pytest.fixture
def cpu_optimizer(monkeypatch):
    from rscotr_amd.optim import FlatAdamW
    monkeypatch.setattr(FlatAdamW, 'launch_step', _torch_step)


def _cfg(tmp_path, **over):
    cfg = dict(optimizer=dict(type='AdamW', lr=1e-2, weight_decay=0.01), optimizer_config=dict(grad_clip=dict(max_norm=0.5, norm_type=2)),
               lr_config=dict(policy='step', step=[4]), runner=dict(type='IterBasedRunner', max_iters=6),
               checkpoint_config=dict(interval=2, max_keep_ckpts=2),
               log_config=dict(interval=3, hooks=[dict(type='TextLoggerHook'), dict(type='TensorboardLoggerHook')]),
               evaluation=dict(interval=3, save_best={'resisc.acc': 1, 'potsdam.miou': 100}, cls=dict(metric='accuracy')),
               work_dir=str(tmp_path), resume_from=None, load_from=None)
    cfg.update(over)
    return cfg


def _run(tmp_path, upto=None, **over):
    from rscotr_amd.runner import build_runner
    torch.manual_seed(0)
    model = Toy()
    lines = []
    runner = build_runner(model, _cfg(tmp_path, **over), Loader(), graph_tasks=(), logger=lines.append, timestamp='t0')
    runner.data_loader.i = runner.iter  # (a resumed run continues the deterministic stream)
    runner.run(upto)
    return runner, model, lines


def test_build_runner_registers_the_configured_hooks(tmp_path, cpu_optimizer):
    from rscotr_amd.hooks import CheckpointHook, TensorboardLoggerHook, TextLoggerHook
    runner, model, lines = _run(tmp_path)
    kinds = [type(h) for h in runner.hooks]
    assert kinds == [CheckpointHook, TextLoggerHook, TensorboardLoggerHook]  # (no loaders given: no evaluation hook)
    assert runner.iter == 6 == runner.max_iters and runner.work_dir == str(tmp_path)
    # checkpoint_config(interval=2, max_keep_ckpts=2): iter_2 was rotated out, latest -> iter_6
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith('iter_')) == ['iter_4.pth', 'iter_6.pth']
    latest = torch.load(os.path.join(tmp_path, 'latest.pth'), weights_only=True)
    assert latest['meta']['iter'] == 6 and set(latest) == {'meta', 'state_dict', 'optimizer'}
    # log_config(interval=3): two reports, each the sample-weighted mean of the keys seen in its window
    recs = [json.loads(l) for l in open(os.path.join(tmp_path, 't0.log.json'))]
    assert [r['iter'] for r in recs] == [3, 6] and all(r['mode'] == 'train' for r in recs)
    rep = [l for l in lines if l.startswith('Iter [')]
    assert len(rep) == 2 and 'lr: ' in rep[0]
    # window 1 = iterations 0, 1, 2: cls batches of 2 and 4 samples, one seg batch of 3
    assert recs[0]['cls.resisc.n'] == pytest.approx((2 * 2 + 4 * 4) / 6, abs=1e-4) and recs[0]['seg.potsdam.n'] == 3
    assert recs[1]['lr'] == pytest.approx(1e-3)  # lr_config step=[4]: decayed in the second window


def test_auto_resume_continues_like_the_uninterrupted_run(tmp_path, cpu_optimizer):
    full, model_full, _ = _run(tmp_path / 'full')
    part, _, _ = _run(tmp_path / 'part', upto=4)
    assert part.iter == 4
    resumed, model_res, lines = _run(tmp_path / 'part', auto_resume=True)
    assert any('resumed from' in l and 'iter 4' in l for l in lines) and resumed.iter == 6
    for (k, a), b in zip(model_full.state_dict().items(), model_res.state_dict().values()):
        assert torch.equal(a, b), k
    assert torch.equal(full.optimizer.flat_m, resumed.optimizer.flat_m) and (full.optimizer.steps == resumed.optimizer.steps).all()
    # resume_from wins over auto_resume; load_from loads weights only (iteration counter and moments start fresh)
    other, model_o, _ = _run(tmp_path / 'o', upto=0, load_from=str(tmp_path / 'full' / 'iter_6.pth'))
    assert other.iter == 0 and float(other.optimizer.flat_m.abs().max()) == 0.0
    for (k, a), b in zip(model_full.state_dict().items(), model_o.state_dict().values()):
        assert torch.equal(a, b), k
    again, _, _ = _run(tmp_path / 'part', upto=4, resume_from=str(tmp_path / 'full' / 'iter_4.pth'), auto_resume=True)
    assert again.iter == 4


def test_evaluation_hook_from_cfg_reports_through_the_logger_and_keeps_the_best(tmp_path, cpu_optimizer):
    from rscotr_amd.engine import MultiDatasetsEvalHook
    from rscotr_amd.runner import build_runner

    class DS:
        def __init__(self, task, key, vals):
            self.task, self.key, self.vals, self.kw = task, key, list(vals), None

        def evaluate(self, results, logger=None, **kw):
            self.kw = kw
            return {self.key: self.vals.pop(0)}

    class DL(list):
        pass
    loaders = {}
    for name, task, key, vals in (('resisc', 'cls', 'acc', [50.0, 40.0]), ('potsdam', 'seg', 'miou', [0.30, 0.45])):
        dl = DL([dict(img=torch.zeros(1, 6))])
        dl.dataset = DS(task, key, vals)
        loaders[name] = dl
    torch.manual_seed(0)
    lines = []
    runner = build_runner(Toy(), _cfg(tmp_path), Loader(), val_dataloaders=loaders, graph_tasks=(), logger=lines.append,
                          timestamp='t1')
    hook = [h for h in runner.hooks if isinstance(h, MultiDatasetsEvalHook)][0]
    hook.test_fn = lambda model, dls: {n: [0] for n in dls}  # (the inference loops: tests/test_engine_*.py)
    assert hook.interval == 3 and not hook.by_epoch and hook.eval_kwargs == dict(cls=dict(metric='accuracy'))
    runner.run()
    assert loaders['resisc'].dataset.kw == dict(metric='accuracy')
    recs = [json.loads(l) for l in open(os.path.join(tmp_path, 't1.log.json'))]
    vals = [r for r in recs if r['mode'] == 'val']
    assert [r['iter'] for r in vals] == [3, 6] and vals[0]['resisc.acc'] == 50.0 and vals[1]['potsdam.miou'] == 0.45
    # weighted mean (50 + 100 * 0.30) / 2 = 40 at iter 3, (40 + 45) / 2 = 42.5 at iter 6: the second one is kept
    best = [f for f in os.listdir(tmp_path) if f.startswith('best_')]
    assert best == ['best_resisc_acc_potsdam_miou_iter_6.pth'] and hook.best_score == pytest.approx(42.5)


def test_unknown_hook_type_and_runner_type_are_refused(tmp_path, cpu_optimizer):
    from rscotr_amd.runner import build_runner
    with pytest.raises(KeyError):
        build_runner(Toy(), _cfg(tmp_path, log_config=dict(interval=1, hooks=[dict(type='WandbLoggerHook')])), Loader(), graph_tasks=())
    with pytest.raises(NotImplementedError):
        build_runner(Toy(), _cfg(tmp_path, runner=dict(type='EpochBasedRunner', max_epochs=1)), Loader(), graph_tasks=())


def test_find_latest_checkpoint(tmp_path):
    from rscotr_amd.hooks import find_latest_checkpoint
    assert find_latest_checkpoint(str(tmp_path)) is None and find_latest_checkpoint(None) is None
    for n in (2, 10, 4):
        open(tmp_path / f'iter_{n}.pth', 'w').close()
    assert find_latest_checkpoint(str(tmp_path)).endswith('iter_10.pth')
    open(tmp_path / 'latest.pth', 'w').close()
    assert find_latest_checkpoint(str(tmp_path)).endswith('latest.pth')


def test_hook_details_follow_mmcv(tmp_path, cpu_optimizer):
    """ADVICE r3 (low), against mmcv 1.6's hooks: (a) a NumPy scalar metric goes into the JSON log; (b) the log window of a key
    is the last n entries OF THAT KEY (LogBuffer.average), also across reports; (c) max_keep_ckpts prunes by iteration number,
    so checkpoints written before a resume go too; (d) time / eta restart at the resumed iteration; (e) the checkpoint's
    meta.epoch is epoch + 1 (IterBasedRunner.save_checkpoint)."""
    import types
    import numpy as np
    from rscotr_amd.hooks import CheckpointHook, TextLoggerHook, _LogHistory
    # (b)
    h = _LogHistory()
    for i, (k, v) in enumerate([('a', 1.0), ('b', 10.0), ('a', 3.0), ('b', 30.0), ('a', 5.0), ('b', 50.0)]):
        h.update({k: v}, 1)
    assert h.average(2) == {'a': 4.0, 'b': 40.0}
    h.update({'a': 7.0}, 1)
    assert h.average(2) == {'a': 6.0, 'b': 40.0}          # 'b' keeps its last two entries across the report
    # (a)
    lines = []
    runner = types.SimpleNamespace(iter=3, max_iters=9, work_dir=str(tmp_path), timestamp='t1', logger=lines.append, epoch=0,
                                   optimizer=types.SimpleNamespace())
    tl = TextLoggerHook(interval=3)
    tl.before_run(runner)
    tl.log(runner, {'dior.bbox_mAP': np.float32(0.25), 'potsdam.mIoU': np.array(0.5), 'note': 'x'}, mode='val')
    rec = json.loads(open(os.path.join(tmp_path, 't1.log.json')).read().strip())
    assert rec['dior.bbox_mAP'] == 0.25 and rec['potsdam.mIoU'] == 0.5 and rec['note'] == 'x'
    # (d)
    runner.iter = 400
    tl.after_resume(runner)
    assert tl.it_last == 400
    # (c) + (e): iter_2 .. iter_6 exist from "before the resume"; the hook of the resumed run writes iter_8 and prunes by name
    run1, _, _ = _run(tmp_path / 'r', upto=6, checkpoint_config=dict(interval=2, max_keep_ckpts=10))
    assert sorted(f for f in os.listdir(tmp_path / 'r') if f.startswith('iter_')) == ['iter_2.pth', 'iter_4.pth', 'iter_6.pth']
    assert torch.load(os.path.join(tmp_path / 'r', 'iter_6.pth'), weights_only=True)['meta']['epoch'] == run1.epoch + 1
    ck = CheckpointHook(interval=2, max_keep_ckpts=2, out_dir=str(tmp_path / 'r'))
    run1.iter = 8
    ck.after_train_iter(run1)
    assert sorted(f for f in os.listdir(tmp_path / 'r') if f.startswith('iter_')) == ['iter_6.pth', 'iter_8.pth']
