"""Host logic of rscotr_amd.engine (mtl/engine/test.py:24-53, mtl/runner/hooks/evaluation.py:29-149): multi-dataset test
dispatch and the evaluation hook's scheduling / weighted best-metric rule, with stand-in model and datasets (no GPU)."""
import os

import pytest
import torch

from rscotr_amd.engine import KeyIndicator, MultiDatasetsEvalHook, collect_results, single_gpu_test


class _DS:
    def __init__(self, task, n, metric):
        self.task, self.n, self.metric, self.seen = task, n, metric, None

    def __len__(self):
        return self.n

    def evaluate(self, results, logger=None, **kw):
        self.seen = (len(results), dict(kw))
        return self.metric(results)


class _Loader(list):
    def __init__(self, dataset, batches):
        super().__init__(batches)
        self.dataset = dataset


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.CLASSES = dict(resisc=('a', 'b'), dior=('x',), potsdam=('p', 'q', 'r'))
        self.calls = []

    def forward(self, task, img, img_metas, return_loss=True, **kw):
        assert not return_loss and not self.training and not torch.is_grad_enabled()
        self.calls.append((task, self.CLASSES, dict(kw)))
        return [float(v) for v in img]  # one result per sample


def _setup(score=lambda r: sum(r)):
    ds = dict(resisc=_DS('cls', 5, lambda r: {'accuracy_top-1': score(r)}),
              dior=_DS('det', 3, lambda r: {'bbox_mAP': 0.5, 'bbox_mAP_50': 0.75}),
              potsdam=_DS('seg', 4, lambda r: {'mIoU': 40.0}))
    mk = lambda t, xs: dict(task=t, img=torch.tensor(xs), img_metas=[{}] * len(xs))
    loaders = dict(resisc=_Loader(ds['resisc'], [mk('cls', [1., 2.]), mk('cls', [3., 4.]), mk('cls', [5.])]),
                   dior=_Loader(ds['dior'], [mk('det', [1., 1., 1.])]),
                   potsdam=_Loader(ds['potsdam'], [mk('seg', [0., 0.]), mk('seg', [0., 0.])]))
    return ds, loaders


def test_single_gpu_test_dispatches_per_task_and_switches_classes():
    ds, loaders = _setup()
    model = _Model().train()
    res = single_gpu_test(model, loaders, kwargs_dict=dict(seg=dict(opacity=0.3)))
    assert {k: len(v) for k, v in res.items()} == dict(resisc=5, dior=3, potsdam=4)
    assert res['resisc'] == [1., 2., 3., 4., 5.]
    by_task = {t: (c, kw) for t, c, kw in model.calls}
    assert by_task['cls'][0] == ('a', 'b') and by_task['det'][0] == ('x',) and by_task['seg'][0] == ('p', 'q', 'r')
    assert by_task['det'][1] == dict(rescale=True)  # mmdet's test loop rescales to the original image
    assert by_task['seg'][1] == {} and by_task['cls'][1] == {}  # (seg display kwargs are consumed by the loop)
    assert isinstance(model.CLASSES, dict) and model.training  # both restored


class _Runner:
    def __init__(self, tmp):
        self.model, self.iter, self.epoch, self.meta, self.work_dir = _Model(), 0, 0, {}, str(tmp)
        self.log_buffer_output, self.log_buffer_ready, self.optimizer, self.logger = {}, False, None, None


def test_eval_hook_weighted_best_metric_and_schedule(tmp_path, monkeypatch):
    ds, loaders = _setup()
    scores = iter([10.0, 30.0, 20.0])
    ds['resisc'].metric = lambda r: {'accuracy_top-1': next(scores)}
    saved = []
    import rscotr_amd.checkpoint as ck
    import copy
    monkeypatch.setattr(ck, 'save_checkpoint', lambda path, model, opt=None, meta=None: (saved.append((path, copy.deepcopy(meta))), open(path, 'w').close()))
    hook = MultiDatasetsEvalHook(loaders, start=None, interval=4, by_epoch=False,
                                 save_best={'resisc.accuracy_top-1': 1, 'dior.bbox_mAP': 100, 'potsdam.mIoU': 1},
                                 det=dict(metric='bbox'), seg=dict(metric='mIoU'), cls=dict(metric='accuracy'))
    r = _Runner(tmp_path)
    hook.before_run(r)
    assert repr(hook.key_indicator) == 'resisc_accuracy_top-1_dior_bbox_mAP_potsdam_mIoU' and hook.out_dir == str(tmp_path)
    evaluated = []
    for it in range(1, 13):  # r.iter counts FINISHED iterations when the hook runs (see _should_evaluate)
        r.iter = it
        before = ds['dior'].seen
        ds['dior'].seen = None
        hook.after_train_iter(r)
        if ds['dior'].seen is not None:
            evaluated.append(it)
        else:
            ds['dior'].seen = before
    assert evaluated == [4, 8, 12]
    assert ds['dior'].seen == (3, dict(metric='bbox')) and ds['potsdam'].seen == (4, dict(metric='mIoU'))
    assert r.log_buffer_output['dior.bbox_mAP_50'] == 0.75 and r.log_buffer_output['eval_iter_num'] == dict(resisc=3, dior=1, potsdam=2)
    # weighted MEAN of the selected metrics: (acc * 1 + 0.5 * 100 + 40 * 1) / 3  (evaluation.py:144-148)
    want = [(10 + 50 + 40) / 3, (30 + 50 + 40) / 3]
    assert [round(m['hook_msgs']['best_score'], 6) for _, m in saved] == [round(w, 6) for w in want]  # third eval is worse
    assert [os.path.basename(p) for p, _ in saved] == ['best_resisc_accuracy_top-1_dior_bbox_mAP_potsdam_mIoU_iter_4.pth',
                                                       'best_resisc_accuracy_top-1_dior_bbox_mAP_potsdam_mIoU_iter_8.pth']
    assert not os.path.exists(saved[0][0]) and os.path.exists(saved[1][0])  # only the best checkpoint is kept
    assert r.meta['hook_msgs']['best_ckpt'] == saved[1][0]


def test_eval_hook_start_and_argument_checks(tmp_path):
    ds, loaders = _setup()
    hook = MultiDatasetsEvalHook(loaders, start=5, interval=3, by_epoch=False, test_fn=lambda m, d: {k: [0] * len(v.dataset) for k, v in d.items()})
    r = _Runner(tmp_path)
    hits = []
    for it in range(1, 15):
        r.iter = it
        ds['dior'].seen = None
        hook.after_train_iter(r)
        if ds['dior'].seen is not None:
            hits.append(it)
    assert hits == [5, 8, 11, 14]
    assert hook.evaluate(r, {k: [0] * 3 for k in loaders}) is None  # no save_best: no key score
    with pytest.raises(ValueError):
        MultiDatasetsEvalHook(loaders, interval=0)
    with pytest.raises(TypeError):
        MultiDatasetsEvalHook([1, 2])
    assert repr(KeyIndicator(**{'a.b': 1, 'c': 2})) == 'a_b_c' and len(KeyIndicator(x=1)) == 1


def test_collect_results_single_process_truncates():
    assert collect_results([1, 2, 3, 4], 3) == [1, 2, 3]
