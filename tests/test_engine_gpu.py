"""rscotr_amd.engine on the real model (GPU): the multi-dataset test dispatch (mtl/engine/test.py:24-53) over
MTL.simple_test_{cls,det,seg} on synthetic datasets, and MultiDatasetsEvalHook (mtl/runner/hooks/evaluation.py) hooked
into the co-training runner: evaluation every `interval` iterations, weighted best-metric checkpoint written in the mmcv
layout and loadable back."""
import numpy as np
import pytest
import torch

from rscotr_amd import synth
from rscotr_amd.engine import MultiDatasetsEvalHook, single_gpu_test
from util import build_model, load_model_cfg

pytestmark = pytest.mark.gpu


class SynthTestSet:
    """A test split: `n` samples of one task, served in batches of `bs`; evaluate() scores what the task's loop returns."""

    def __init__(self, task, n, bs, device, seed):
        self.task, self.n = task, n
        self.batches = []
        for i in range(0, n, bs):
            b = synth.make_batch(task, min(bs, n - i), 64, seed=seed + i, device=device)
            self.batches.append(dict(task=task, img=b['img'], img_metas=b['img_metas']))
        self.labels = np.arange(n) % 3

    def __len__(self):
        return self.n

    def evaluate(self, results, logger=None, **kw):
        assert len(results) == self.n
        if self.task == 'cls':
            assert results[0].shape == (45,)
            return {'accuracy_top-1': float(np.mean([int(np.argmax(r)) == l for r, l in zip(results, self.labels)]) * 100)}
        if self.task == 'det':
            assert len(results[0]) == 20 and results[0][0].shape[1] == 5
            return {'bbox_mAP': float(np.mean([np.concatenate(r)[:, 4].mean() for r in results]))}
        assert results[0].shape == (64, 64)
        return {'mIoU': float(np.mean([(r == 0).mean() for r in results]) * 100)}


class _Loader(list):
    def __init__(self, ds):
        super().__init__(ds.batches)
        self.dataset = ds


def test_multi_dataset_test_and_eval_hook(cuda, tmp_path):
    from rscotr_amd.checkpoint import load_checkpoint
    from rscotr_amd.data import build_synthetic_multidataloader
    from rscotr_amd.runner import build_runner
    cfg, mcfg = load_model_cfg(tiny=True)
    mcfg['test_cfg']['det']['max_per_img'] = 10
    model = build_model(mcfg).to(cuda)
    model.CLASSES = dict(resisc=tuple(range(45)), dior=tuple(range(20)), potsdam=tuple(range(5)))
    loaders = dict(resisc=_Loader(SynthTestSet('cls', 5, 2, cuda, 100)), dior=_Loader(SynthTestSet('det', 3, 2, cuda, 200)),
                   potsdam=_Loader(SynthTestSet('seg', 4, 2, cuda, 300)))
    res = single_gpu_test(model, loaders)
    assert {k: len(v) for k, v in res.items()} == dict(resisc=5, dior=3, potsdam=4) and model.training
    # the same per-sample outputs as calling the inference path batch by batch
    model.eval()
    direct = model(return_loss=False, **loaders['resisc'][0])
    model.train()
    assert np.allclose(np.stack(res['resisc'][:2]), np.stack(direct), rtol=1e-5, atol=1e-7)

    train_loader = build_synthetic_multidataloader(cfg, cuda, size=64, batch_size=2)
    runner = build_runner(model, cfg, train_loader, graph_tasks=())
    runner.work_dir = str(tmp_path)
    hook = MultiDatasetsEvalHook(loaders, interval=3, by_epoch=False,
                                 save_best={'resisc.accuracy_top-1': 1, 'dior.bbox_mAP': 100, 'potsdam.mIoU': 0.5})
    runner.register_hook(hook)
    runner.run(6)
    assert runner.iter == 6 and runner.log_buffer_ready
    out = runner.log_buffer_output
    assert set(out) >= {'resisc.accuracy_top-1', 'dior.bbox_mAP', 'potsdam.mIoU', 'eval_iter_num'}
    want = (out['resisc.accuracy_top-1'] + 100 * out['dior.bbox_mAP'] + 0.5 * out['potsdam.mIoU']) / 3
    assert hook.best_score >= want - 1e-9 and hook.best_ckpt_path is not None
    fresh = build_model(mcfg, seed=9).to(cuda)
    load_checkpoint(fresh, hook.best_ckpt_path)
    sd, fd = model.state_dict(), fresh.state_dict()
    if hook.best_ckpt_path.endswith('iter_6.pth'):  # the best evaluation was the last: weights equal the live model's
        assert all(torch.equal(sd[k], fd[k]) for k in sd)
    runner.optimizer.close()
