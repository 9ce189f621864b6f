"""Evaluation side of the co-training loop: the multi-dataset test dispatch of
`mtl/engine/test.py:24-53` and the evaluation hook of `mtl/runner/hooks/evaluation.py:29-149`, over the
`MTL.simple_test_{cls,det,seg}` inference path (rscotr_amd/mtl.py).

`single_gpu_test(model, data_loaders, show, out_dir, kwargs_dict)` walks every dataset's loader with the test loop of
its task (what the reference borrows from mmcls / mmdet / mmseg `apis.single_gpu_test`: `model(return_loss=False,
**data)` per batch, results concatenated in dataset order; det passes `rescale=True`) after switching `model.CLASSES`
to the dataset's own classes, and returns `{dataset_name: results}`; `multi_gpu_test` runs the same loops on every
rank's shard and gathers the shards in the interleaved order of mmcv's `collect_results_gpu`.

`MultiDatasetsEvalHook` evaluates every `interval` iterations (or epochs) from `start` on, calls each dataset's
`evaluate(results, logger=..., **eval_kwargs[task])`, publishes `'{dataset}.{metric}'` values to the runner's log
buffer, and — with `save_best` (a key, a list of keys or a {key: weight} dict) — keeps the checkpoint whose WEIGHTED
MEAN of the selected metrics (`evaluation.py:144-148`: sum(metric * weight) / number of keys, rule 'greater') is best.
"""
import os
from collections import OrderedDict

import torch


def _loop(model, loader, **kwargs):
    results = []
    was_training = model.training
    model.eval()
    try:
        for data in loader:
            with torch.no_grad():
                result = model(return_loss=False, **dict(data, **kwargs))
            results.extend(result if isinstance(result, (list, tuple)) else [result])
    finally:
        model.train(was_training)
    return results


def _test_cls(model, loader, show=False, out_dir=None, **kwargs):
    """mmcls.apis.single_gpu_test: per-sample class-score vectors."""
    return _loop(model, loader, **kwargs)


def _test_det(model, loader, show=False, out_dir=None, show_score_thr=0.3, **kwargs):
    """mmdet.apis.single_gpu_test: per-image lists of per-class (k, 5) arrays, boxes in original-image coordinates."""
    return _loop(model, loader, rescale=True, **kwargs)


def _test_seg(model, loader, show=False, out_dir=None, efficient_test=False, opacity=0.5, pre_eval=False,
              format_only=False, format_args=None, **kwargs):
    """mmseg.apis.single_gpu_test (plain mode): per-image label maps at the original size."""
    return _loop(model, loader, **kwargs)


single_gpu_single_dataset_test = dict(cv=_loop, cls=_test_cls, det=_test_det, seg=_test_seg)


def single_gpu_test(model, data_loaders, show=False, out_dir=None, kwargs_dict=None):
    """mtl/engine/test.py:24-39."""
    results = dict()
    CLASSES = model.CLASSES
    kwargs_dict = dict() if kwargs_dict is None else kwargs_dict
    try:
        for name, dataloader in data_loaders.items():
            task = getattr(dataloader.dataset, 'task', 'cv')
            kwargs = kwargs_dict.get(task, dict())
            if CLASSES is not None:
                model.CLASSES = CLASSES[name]
            results[name] = single_gpu_single_dataset_test[task](model, dataloader, show, out_dir, **kwargs)
    finally:
        model.CLASSES = CLASSES
    return results


def collect_results(result_part, size):
    """mmcv.engine.collect_results_gpu: every rank's shard gathered on rank 0 and interleaved (sample i of the dataset
    sits at position i // world of rank i % world with a DistributedSampler), truncated to the dataset size."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return result_part[:size]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, result_part)
    if dist.get_rank() != 0:
        return None
    ordered = []
    for res in zip(*parts):
        ordered.extend(list(res))
    # shards may differ in length by one when the sampler does not pad
    longest = max(len(p) for p in parts)
    for i in range(min(len(p) for p in parts), longest):
        ordered.extend(p[i] for p in parts if len(p) > i)
    return ordered[:size]


def multi_gpu_test(model, data_loaders, tmpdir=None, gpu_collect=False, kwargs_dict=None):
    """mtl/engine/test.py:42-53: results of rank 0 are the whole datasets', None elsewhere."""
    results = dict()
    kwargs_dict = dict() if kwargs_dict is None else kwargs_dict
    for name, dataloader in data_loaders.items():
        task = getattr(dataloader.dataset, 'task', 'cv')
        kwargs = {k: v for k, v in kwargs_dict.get(task, dict()).items()}
        part = single_gpu_single_dataset_test[task](model, dataloader, False, None, **kwargs)
        results[name] = collect_results(part, len(dataloader.dataset))
    return results


class KeyIndicator:
    """`evaluation.py:9-26`: the {metric key: weight} selection behind `save_best`; its repr names the checkpoint."""

    def __init__(self, **kwargs):
        self.key_indicator = dict(**kwargs)

    def __getitem__(self, item):
        return self.key_indicator[item]

    def __repr__(self):
        return '_'.join(key.replace('.', '_') for key in self.key_indicator)

    def __len__(self):
        return len(self.key_indicator)

    def items(self):
        return self.key_indicator.items()


class MultiDatasetsEvalHook:
    """`mtl/runner/hooks/evaluation.py:29-149` on rscotr_amd.runner.IterBasedRunner (hook point: after_train_iter)."""

    rule_map = {'greater': lambda x, y: x > y, 'less': lambda x, y: x < y}
    init_value_map = {'greater': -float('inf'), 'less': float('inf')}

    def __init__(self, dataloaders, start=None, interval=1, by_epoch=True, save_best=None, test_fn=None,
                 greater_keys=None, less_keys=None, out_dir=None, file_client_args=None, **eval_kwargs):
        if not isinstance(dataloaders, dict):
            raise TypeError(f'dataloaders must be a dict of loaders, but got {type(dataloaders)}')
        if interval <= 0:
            raise ValueError(f'interval must be a positive number, but got {interval}')
        assert isinstance(by_epoch, bool), '``by_epoch`` should be a boolean'
        if start is not None and start < 0:
            raise ValueError(f'The evaluation start epoch {start} is smaller than 0')
        self.dataloaders, self.interval, self.start, self.by_epoch = dataloaders, interval, start, by_epoch
        assert isinstance(save_best, (str, list, dict)) or save_best is None, \
            f'"save_best" should be a str, or list, or dict, or None rather than {type(save_best)}'
        if isinstance(save_best, str):
            save_best = {save_best: 1}
        if isinstance(save_best, list):
            save_best = {key: 1 for key in save_best}
        self.save_best = save_best
        self.eval_kwargs = eval_kwargs
        self.initial_flag = True
        self.test_fn = single_gpu_test if test_fn is None else test_fn
        self.greater_keys, self.less_keys = greater_keys, less_keys
        self.best_ckpt_path, self.best_score = None, None
        if self.save_best is not None:
            self.rule = 'greater'  # the reference fixes the rule (`_init_rule('greater', save_best)`, :107)
            self.key_indicator = KeyIndicator(**self.save_best)
            self.compare_func = self.rule_map[self.rule]
            self.best_score = self.init_value_map[self.rule]
        self.out_dir = out_dir

    # ---- mmcv EvalHook scheduling --------------------------------------------------------------------------------
    def _should_evaluate(self, runner):
        """mmcv EvalHook._should_evaluate.  mmcv calls the hook BEFORE its runner increments the counter and tests
        `counter + 1`; this repo's runner has already counted the finished iteration, so `done` is that same number."""
        done = runner.epoch + 1 if self.by_epoch else runner.iter
        if self.start is None:
            return done % self.interval == 0
        if done < self.start:
            return False
        return (done - self.start) % self.interval == 0

    def before_run(self, runner):
        if self.out_dir is None:
            self.out_dir = getattr(runner, 'work_dir', None)

    def before_train_iter(self, runner):
        """Evaluate the resumed / initial model once when training starts at or after `start` (mmcv EvalHook)."""
        if self.by_epoch or not self.initial_flag:
            return
        if self.start is not None and runner.iter >= self.start:
            self._do_evaluate(runner)
        self.initial_flag = False

    def after_train_iter(self, runner):
        if not self.by_epoch and self._should_evaluate(runner):
            self._do_evaluate(runner)

    def after_train_epoch(self, runner):
        if self.by_epoch and self._should_evaluate(runner):
            self._do_evaluate(runner)

    # ---- evaluation.py:118-148 -----------------------------------------------------------------------------------
    def _do_evaluate(self, runner):
        results_dict = self.test_fn(runner.model, self.dataloaders)
        runner.log_buffer_output['eval_iter_num'] = {name: len(dl) for name, dl in self.dataloaders.items()}
        key_score = self.evaluate(runner, results_dict)
        # the key_score may be `None` (or 0) so it needs to skip the action to save the best checkpoint
        if self.save_best and key_score:
            self._save_ckpt(runner, key_score)

    def evaluate(self, runner, results_dict):
        eval_res = OrderedDict()
        for dataset_name, dataloader in self.dataloaders.items():
            task = getattr(dataloader.dataset, 'task')
            metrics = dataloader.dataset.evaluate(results_dict[dataset_name], logger=getattr(runner, 'logger', None),
                                                  **(self.eval_kwargs.get(task, None) or {}))
            eval_res.update({f'{dataset_name}.{metric_name}': val for metric_name, val in metrics.items()})
        for name, val in eval_res.items():
            runner.log_buffer_output[name] = val
        runner.log_buffer_ready = True
        if self.save_best is not None:
            metrics_sum = sum(eval_res.get(key, 0.) * weight for key, weight in self.key_indicator.items())
            return metrics_sum / len(self.key_indicator)
        return None

    def _save_ckpt(self, runner, key_score):
        """mmcv EvalHook._save_ckpt: keep only the best checkpoint, named after the selected keys."""
        if not self.compare_func(key_score, self.best_score):
            return
        self.best_score = key_score
        runner.meta = getattr(runner, 'meta', None) or {}
        runner.meta.setdefault('hook_msgs', {})['best_score'] = key_score
        if self.out_dir is None:
            return
        from .checkpoint import save_checkpoint
        cur_type, cur_time = ('epoch', runner.epoch + 1) if self.by_epoch else ('iter', runner.iter)
        if self.best_ckpt_path and os.path.isfile(self.best_ckpt_path):
            os.remove(self.best_ckpt_path)
        self.best_ckpt_path = os.path.join(self.out_dir, f'best_{self.key_indicator!r}_{cur_type}_{cur_time}.pth')
        runner.meta['hook_msgs']['best_ckpt'] = self.best_ckpt_path
        os.makedirs(self.out_dir, exist_ok=True)
        save_checkpoint(self.best_ckpt_path, runner.model, getattr(runner, 'optimizer', None),
                        meta=dict(runner.meta, iter=runner.iter, epoch=getattr(runner, 'epoch', 0)))
