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


def _toy_datasets(root, rng):
    """Three on-disk toy test splits in the layouts the repo's readers take (rscotr_amd/pipeline.py): an image-folder
    classification set, a COCO-json detection set, a Potsdam-style tile set."""
    import json
    from PIL import Image
    from rscotr_amd.pipeline import CocoDetDataset, FolderClsDataset, TileSegDataset
    img = lambda h, w: Image.fromarray(rng.randint(0, 255, size=(h, w, 3)).astype(np.uint8))
    for c in [f'class{i:02d}' for i in range(45)][:4]:
        (root / 'cls' / c).mkdir(parents=True)
        for i in range(2):
            img(64, 64).save(root / 'cls' / c / f'{i}.png')
    (root / 'det').mkdir()
    names = [f'c{i}' for i in range(20)]
    images, anns = [], []
    for i in range(3):
        img(64, 64).save(root / 'det' / f'{i}.png')
        images.append(dict(id=i, file_name=f'{i}.png', width=64, height=64))
        for j in range(2):
            anns.append(dict(id=len(anns), image_id=i, category_id=1 + (3 * i + j) % 20, bbox=[4 + 24 * j, 8, 20, 30 + 4 * i],
                             area=20 * (30 + 4 * i), iscrowd=0))
    (root / 'det.json').write_text(json.dumps(dict(images=images, annotations=anns,
                                                   categories=[dict(id=k + 1, name=n) for k, n in enumerate(names)])))
    (root / 'seg_img').mkdir(); (root / 'seg_ann').mkdir()
    for i in range(4):
        img(64, 64).save(root / 'seg_img' / f't{i}.png')
        Image.fromarray(rng.randint(0, 7, size=(64, 64)).astype(np.uint8)).save(root / 'seg_ann' / f't{i}.png')
    return (FolderClsDataset(str(root / 'cls')), CocoDetDataset(str(root / 'det.json'), str(root / 'det'), classes=names),
            TileSegDataset(str(root / 'seg_img'), str(root / 'seg_ann'), ignore_index=5))


def test_eval_hook_on_the_repos_own_datasets(cuda, tmp_path):
    """MultiDatasetsEvalHook end to end on on-disk datasets read by the repo's own readers, scored by their
    own evaluate() with the evaluation kwargs of the reference config (configs/multi/MTL_slvlcls_...potsdam.py:222-237) — no stub
    dataset in the loop; the metrics equal what rscotr_amd.metrics computes from the raw test-loop results."""
    from rscotr_amd.data import build_synthetic_multidataloader
    from rscotr_amd.metrics import accuracy, coco_bbox_map
    from rscotr_amd.pipeline import DeviceCollate, DeviceLoader
    from rscotr_amd.runner import build_runner
    cfg, mcfg = load_model_cfg(tiny=True)
    mcfg['test_cfg']['det']['max_per_img'] = 10
    model = build_model(mcfg).to(cuda)
    cls_ds, det_ds, seg_ds = _toy_datasets(tmp_path, np.random.RandomState(7))
    model.CLASSES = dict(resisc=tuple(range(45)), dior=det_ds.CLASSES, potsdam=seg_ds.CLASSES[:5])
    test_collate = lambda task, **kw: DeviceCollate(task, cuda, flip_prob=0.0, **kw)
    loaders = dict(resisc=DeviceLoader(cls_ds, test_collate('cls'), 3, test_mode=True),
                   dior=DeviceLoader(det_ds, test_collate('det', size_divisor=32), 2, test_mode=True),
                   potsdam=DeviceLoader(seg_ds, test_collate('seg'), 2, test_mode=True))
    res = single_gpu_test(model, loaders)
    assert {k: len(v) for k, v in res.items()} == dict(resisc=8, dior=3, potsdam=4)
    eval_kwargs = dict(cls=dict(metric='accuracy'), det=dict(metric='bbox', iou_thrs=[0.5], classwise=True),
                       seg=dict(metric=['mFscore', 'mIoU'], pre_eval=True, classwise=True))
    runner = build_runner(model, cfg, build_synthetic_multidataloader(cfg, cuda, size=64, batch_size=2), graph_tasks=())
    runner.work_dir = str(tmp_path)
    hook = MultiDatasetsEvalHook(loaders, interval=2, by_epoch=False,
                                 save_best={'resisc.accuracy_top-1': 1, 'dior.bbox_mAP': 100, 'potsdam.mFscore': 100}, **eval_kwargs)
    runner.register_hook(hook)
    runner.run(2)
    out = runner.log_buffer_output
    assert set(out) >= {'resisc.accuracy_top-1', 'resisc.accuracy_top-5', 'dior.bbox_mAP', 'dior.bbox_mAP_50', 'potsdam.mFscore',
                        'potsdam.mIoU', 'potsdam.aAcc', 'potsdam.IoU.building'}
    assert 0.0 <= out['resisc.accuracy_top-1'] <= out['resisc.accuracy_top-5'] <= 100.0
    assert -1.0 <= out['dior.bbox_mAP'] <= 1.0 and 0.0 <= out['potsdam.aAcc'] <= 1.0
    # the hook's numbers are the metric functions applied to what the test loop returns for the weights after two iterations
    res2 = single_gpu_test(model, loaders)
    assert out['resisc.accuracy_top-1'] == pytest.approx(accuracy(res2['resisc'], [l for _, l in cls_ds.items], topk=(1,))['accuracy_top-1'])
    want = coco_bbox_map(res2['dior'], [it[1] for it in det_ds.items], [it[2] for it in det_ds.items], det_ds.CLASSES, iou_thrs=[0.5])
    assert out['dior.bbox_mAP'] == want['bbox_mAP']
    runner.optimizer.close()
