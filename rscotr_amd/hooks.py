"""The training hooks `runner.register_training_hooks(lr_config, optimizer_config, checkpoint_config, log_config)` installs
in the reference (`mtl/apis/train.py:77-83`, mmcv 1.6 `BaseRunner.register_training_hooks`), as far as the co-training
loop of this repo needs them: `CheckpointHook`, `TextLoggerHook`, `TensorboardLoggerHook` — the type strings of
`configs/multi/default_runtime.py` — plus `find_latest_checkpoint` (mmdet.utils, `train.py:109-113`).  The LR schedule and the
optimizer hook live in the runner itself (`rscotr_amd/runner.py`, `rscotr_amd/optim.py`); the evaluation hook in
`rscotr_amd/engine.py`.

Hook protocol (the subset of mmcv's `Hook` this runner calls): `before_run(runner)`, `before_train_iter(runner)`,
`after_train_iter(runner)`, `after_run(runner)`; `runner.iter` has already counted the finished iteration when
`after_train_iter` runs, so "every n iterations" is `runner.iter % n == 0` (mmcv tests `(runner.iter + 1) % n` BEFORE its
runner increments the counter: the same iterations)."""
import glob
import json
import os
import re
import time
from collections import OrderedDict

import torch

HOOKS = {}


def register_hook(cls):
    HOOKS[cls.__name__] = cls
    return cls


def build_hook(cfg, **defaults):
    cfg = dict(cfg)
    typ = cfg.pop('type')
    if typ not in HOOKS:
        raise KeyError(f'{typ} is not a registered hook (known: {sorted(HOOKS)})')
    return HOOKS[typ](**dict(defaults, **cfg))


@register_hook
class CheckpointHook:
    """mmcv CheckpointHook on an iteration-based runner: `iter_{n}.pth` every `interval` iterations (and after the last one
    with `save_last`), `latest.pth` pointing at the newest, at most `max_keep_ckpts` kept (cfg `checkpoint_config`,
    `...potsdam.py:218`)."""

    def __init__(self, interval=-1, by_epoch=False, save_optimizer=True, out_dir=None, max_keep_ckpts=-1, save_last=True,
                 **kwargs):
        self.interval, self.by_epoch, self.save_optimizer = interval, by_epoch, save_optimizer
        self.out_dir, self.max_keep_ckpts, self.save_last = out_dir, max_keep_ckpts, save_last
        self.saved = []

    def before_run(self, runner):
        if self.out_dir is None:
            self.out_dir = getattr(runner, 'work_dir', None)

    def _due(self, runner):
        if self.by_epoch or self.out_dir is None:
            return False
        if self.interval > 0 and runner.iter % self.interval == 0:
            return True
        return bool(self.save_last and getattr(runner, 'max_iters', None) and runner.iter == runner.max_iters)

    def after_train_iter(self, runner):
        if not self._due(runner):
            return
        from .checkpoint import save_checkpoint
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, f'iter_{runner.iter}.pth')
        # (mmcv IterBasedRunner.save_checkpoint stores `epoch + 1` and `iter`; its resume reads both back as they are)
        meta = dict(getattr(runner, 'meta', None) or {}, iter=runner.iter, epoch=getattr(runner, 'epoch', 0) + 1)
        if _is_rank0():
            save_checkpoint(path, runner.model, runner.optimizer if self.save_optimizer else None, meta=meta)
            latest = os.path.join(self.out_dir, 'latest.pth')
            if os.path.lexists(latest):
                os.remove(latest)
            try:
                os.symlink(os.path.basename(path), latest)
            except OSError:  # (file systems without symlinks: mmcv copies)
                import shutil
                shutil.copy(path, latest)
            self.saved.append(path)
            if self.max_keep_ckpts > 0 and self.interval > 0:
                # mmcv derives the redundant names from the iteration number (so checkpoints written before a resume are
                # pruned too): iter_{it - k * interval}.pth for k = max_keep_ckpts, max_keep_ckpts + 1, ... until one is missing
                for step in range(runner.iter - self.max_keep_ckpts * self.interval, 0, -self.interval):
                    old = os.path.join(self.out_dir, f'iter_{step}.pth')
                    if not os.path.isfile(old):
                        break
                    os.remove(old)
        runner.meta = getattr(runner, 'meta', None) or {}
        runner.meta.setdefault('hook_msgs', {})['last_ckpt'] = path


class _LogHistory:
    """mmcv LogBuffer: per key the history of (value, sample count); `average(n)` = sample-weighted mean of the last n
    entries of every key.  Values stay device-side (LazyLogVars) until an average is asked for."""

    def __init__(self):
        self.pending = []          # (log_vars, count) of the iterations since the last report: read back (one device ->
        self.hist = OrderedDict()  # host copy each) only when a report is due; per key the (value, count) history

    def update(self, log_vars, count):
        self.pending.append((log_vars, count))

    def average(self, n):
        """mmcv LogBuffer.average(n): per KEY the sample-weighted mean of its last n entries — with the tasks alternating a
        key appears once per round, so its window reaches back n of ITS iterations, across reports (the history is kept)."""
        for lv, c in self.pending:
            for k in lv:
                self.hist.setdefault(k, []).append((float(lv[k]), c))
        self.pending = []
        out = OrderedDict()
        for k, vs in self.hist.items():
            if n > 0 and len(vs) > n:
                del vs[:-n]      # (nothing older than the window is ever read again)
            tot = sum(c for _, c in vs)
            out[k] = sum(v * c for v, c in vs) / max(tot, 1)
        return out


class _LoggerHook:
    """mmcv LoggerHook: collects the log variables of every iteration and reports their average every `interval`."""

    def __init__(self, interval=10, ignore_last=True, reset_flag=False, by_epoch=False, **kwargs):
        self.interval, self.ignore_last, self.by_epoch = interval, ignore_last, by_epoch
        self.history = _LogHistory()
        self.t_last, self.it_last = None, 0

    def before_run(self, runner):
        self.t_last, self.it_last = time.time(), runner.iter

    def after_train_iter(self, runner):
        out = getattr(runner, 'outputs', None)
        if out is not None and out.get('log_vars') is not None:
            self.history.update(out['log_vars'], out.get('num_samples', 1))
        ready = bool(getattr(runner, 'log_buffer_ready', False))
        if runner.iter % self.interval == 0 or ready:
            tags = self.history.average(self.interval)
            if ready:  # the evaluation hook published '{dataset}.{metric}' values
                val = OrderedDict(runner.log_buffer_output)
                # every logger hook reports them; the LAST one clears the buffer (mmcv: `reset_flag` of the last LoggerHook)
                loggers = [h for h in getattr(runner, 'hooks', []) if isinstance(h, _LoggerHook)]
                if not loggers or loggers[-1] is self:
                    runner.log_buffer_output.clear()
                    runner.log_buffer_ready = False
                self.log(runner, val, mode='val')
            if tags:
                now = time.time()
                n = max(runner.iter - self.it_last, 1)
                tags['time'] = (now - (self.t_last or now)) / n
                self.t_last, self.it_last = now, runner.iter
                self.log(runner, tags, mode='train')

    def after_resume(self, runner):
        """The iteration counter jumped (runner.resume): time / eta are measured from here."""
        self.t_last, self.it_last = time.time(), runner.iter

    def after_run(self, runner):
        pass

    def log(self, runner, tags, mode):
        raise NotImplementedError


def _is_rank0():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


@register_hook
class TextLoggerHook(_LoggerHook):
    """mmcv TextLoggerHook: one line per report through the runner's logger and one JSON object per report appended to
    `{work_dir}/{timestamp}.log.json` (`log_config`, `...potsdam.py:219`, `default_runtime.py`)."""

    def __init__(self, by_epoch=False, interval=10, ignore_last=True, reset_flag=False, interval_exp_name=1000, out_dir=None,
                 out_suffix=('.log.json', '.log', '.py'), keep_local=True, file_client_args=None, **kwargs):
        super().__init__(interval=interval, ignore_last=ignore_last, reset_flag=reset_flag, by_epoch=by_epoch)
        self.out_dir, self.json_path = out_dir, None

    def before_run(self, runner):
        super().before_run(runner)
        d = self.out_dir or getattr(runner, 'work_dir', None)
        if d is not None and _is_rank0():
            os.makedirs(d, exist_ok=True)
            stamp = getattr(runner, 'timestamp', None) or time.strftime('%Y%m%d_%H%M%S', time.localtime())
            self.json_path = os.path.join(d, f'{stamp}.log.json')

    def log(self, runner, tags, mode):
        if not _is_rank0():
            return
        # mmcv logs `runner.current_lr()[0]`: the first parameter group's rate (here: one group per parameter, the first is
        # backbone.patch_embed.projection.weight with its lr_mult)
        lr = float(runner.optimizer.base_lr[0]) * runner.optimizer.lr_factor if hasattr(runner.optimizer, 'base_lr') else 0.0
        rec = OrderedDict(mode=mode, epoch=getattr(runner, 'epoch', 0) + 1, iter=runner.iter)
        if mode == 'train':
            mem = int(torch.cuda.max_memory_allocated() / (1024 * 1024)) if torch.cuda.is_available() else 0
            rec.update(lr=float(lr), memory=mem)
            head = f'Iter [{runner.iter}/{getattr(runner, "max_iters", None) or "?"}]\tlr: {lr:.3e}, '
            t = tags.get('time')
            if t is not None and getattr(runner, 'max_iters', None):
                eta = int(t * (runner.max_iters - runner.iter))
                head += f'eta: {eta // 3600}:{eta % 3600 // 60:02d}:{eta % 60:02d}, time: {t:.3f}, memory: {mem}, '
        else:
            head = f'Iter({mode}) [{runner.iter}]\t'
        items = []
        for k, v in tags.items():
            if hasattr(v, 'item'):  # NumPy scalars / 0-d arrays / one-element tensors returned by dataset.evaluate()
                try:
                    v = v.item()
                except (ValueError, RuntimeError):
                    v = str(v)
            rec[k] = round(float(v), 5) if isinstance(v, (int, float)) else v
            if k != 'time':
                items.append(f'{k}: {v:.4f}' if isinstance(v, float) else f'{k}: {v}')
        runner.logger(head + ', '.join(items))
        if self.json_path is not None:
            with open(self.json_path, 'a') as fh:
                json.dump(rec, fh)
                fh.write('\n')


@register_hook
class TensorboardLoggerHook(_LoggerHook):
    """mmcv TensorboardLoggerHook (`default_runtime.py`).  With `torch.utils.tensorboard` importable the scalars go to
    `{work_dir}/tf_logs`; where the `tensorboard` package is absent (this image) the hook is an explicit no-op that says so
    once — the configs name it, so it must build."""

    def __init__(self, log_dir=None, interval=10, ignore_last=True, reset_flag=False, by_epoch=False, **kwargs):
        super().__init__(interval=interval, ignore_last=ignore_last, reset_flag=reset_flag, by_epoch=by_epoch)
        self.log_dir, self.writer, self.active = log_dir, None, None

    def before_run(self, runner):
        super().before_run(runner)
        d = self.log_dir or (os.path.join(runner.work_dir, 'tf_logs') if getattr(runner, 'work_dir', None) else None)
        self.active = False
        if d is None or not _is_rank0():
            return
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer, self.active = SummaryWriter(d), True
        except Exception as e:  # noqa: BLE001 — ImportError, or the package's own import-time failure
            runner.logger(f'TensorboardLoggerHook: tensorboard is not available ({type(e).__name__}); scalars are not written')

    def log(self, runner, tags, mode):
        if self.writer is None:
            return
        for k, v in tags.items():
            if isinstance(v, (int, float)):
                self.writer.add_scalar(f'{mode}/{k}', v, runner.iter)

    def after_run(self, runner):
        if self.writer is not None:
            self.writer.close()


def find_latest_checkpoint(path, suffix='pth'):
    """mmdet.utils.find_latest_checkpoint: `latest.pth` if present, else the `*.pth` with the largest trailing number."""
    if path is None or not os.path.isdir(path):
        return None
    latest = os.path.join(path, f'latest.{suffix}')
    if os.path.exists(latest):
        return latest
    best, best_n = None, -1
    for f in glob.glob(os.path.join(path, f'*.{suffix}')):
        m = re.search(r'_(\d+)\.' + suffix + '$', os.path.basename(f))
        if m and int(m.group(1)) > best_n:
            best, best_n = f, int(m.group(1))
    return best
