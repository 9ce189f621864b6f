"""The co-training loop: what mmcv's IterBasedRunner + OptimizerHook do around
`MTL.train_step` in the reference (mtl/apis/train.py:24-120).

Per iteration: batch = next(MultiDataLoader) -> model.train_step -> zero_grad -> backward
[gradient buckets all-reduced while backward runs] -> clip_grad_norm_ -> AdamW.step -> LR
schedule / logging.  Hook order is the reference's: zero_grad, backward, clip, step.
"""
import contextlib
import os
import time
from collections import OrderedDict

import torch

from .dist import GradSync, is_dist
from .mtl import LazyLogVars
from . import ops
from .optim import StepLrUpdater, build_optimizer


_HOST_TIMES = os.environ.get('RSCOTR_HOST_TIMES') == '1'


class GraphedTask:
    """One task's whole iteration — forward, loss, zero_grad, backward, clip, AdamW — captured once
    into a hipGraph and replayed (the step is launch-bound: ~2-3k kernel launches per iteration).

    Every task is shape-static: cls (the Mixup/CutMix draw is turned into three small device tensors,
    rscotr_amd.cls_head.Augments.apply_static), seg, and det in its DetStatic form (ground truth padded
    to a capacity, extra masked denoising slots, assignment solved on the device: rscotr_amd.det_head).
    Per replay the host copies the batch into the static inputs, refreshes the augment parameters and
    the optimizer's per-tensor table (pinned memory read by a captured H2D copy), launches the graph and
    reads the packed loss vector back (the step's one device->host copy)."""

    TENSOR_KEYS = ('img', 'gt_label', 'gt_semantic_seg')

    def __init__(self, runner, task, batch):
        self.runner, self.task = runner, task
        self.model, self.opt = runner.model, runner.optimizer
        self.static = {k: batch[k].clone() for k in self.TENSOR_KEYS if k in batch}
        self.meta = {k: v for k, v in batch.items() if k not in self.static}
        self.aug = None
        self.det_static = None
        if task == 'det':
            from .det_head import DetStatic, _round_up
            gcap = None
            if runner.sync is not None:
                # one capacity for all ranks (MAX of what each rank's capture batch needs, with headroom): the ranks then
                # take the same graph-or-eager decision for most batches; when they do not, both paths issue the same
                # collective sequence (see IterBasedRunner._train_iter)
                import torch.distributed as dist
                need = torch.tensor([max([int(l.shape[0]) for l in batch['gt_labels']] + [1])], device=batch['img'].device)
                dist.all_reduce(need, op=dist.ReduceOp.MAX)
                gcap = _round_up(max(int(need.item()), 32), 32)
            self.det_static = DetStatic(self.model.bbox_head, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'],
                                        batch['img'].device, gcap=gcap)
            # pinned staging block of the later batches (DetStatic packs a batch into one block when the loader left host
            # copies of the ground truth: one upload per det iteration)
            self.det_pinned = None
            if self.det_static.host_blob is not None:
                self.det_pinned = torch.empty(self.det_static.host_blob.numel(), dtype=torch.uint8).pin_memory()
        if task == 'cls':
            sp = self.model.cls_augments.static_params(self._draw(), batch['img'].shape[0])
            self.aug = {k: v.to(batch['img'].device) for k, v in sp.items()}
            self.aug_host = {k: torch.empty_like(v).pin_memory() for k, v in sp.items()}
        self.names = None
        self.packed = None
        self.done = None
        self.weight = self.model.task_weight[task]
        self.table = self.opt.new_host_table()  # this graph's own pinned optimizer table
        # Distributed, two forms.  `exchange_in_graph` (default): the whole iteration INCLUDING the RCCL collectives is one
        # hipGraph — backward launches each gradient bucket's all-reduce from its hooks as soon as the bucket is complete
        # (torch captures the collective on RCCL's stream with event edges to and from the compute stream), so in the
        # replayed graph the exchange of bucket k runs under the backward kernels of bucket k+1, and clip + AdamW follow the
        # last wait inside the graph.  `split` (RSCOTR_DIST_CAPTURE=0, or if capturing the collectives fails): the graph
        # holds forward + backward only; buckets, log vector and optimizer are issued eagerly after each replay.
        self.exchange_in_graph = runner.sync is not None and os.environ.get('RSCOTR_DIST_CAPTURE', '1') != '0'
        self.split = runner.sync is not None and not self.exchange_in_graph
        try:
            self._warm_and_capture()
        except Exception as e:  # noqa: BLE001 — any failure of the collective capture: fall back to the split form
            if not self.exchange_in_graph:
                raise
            import warnings
            warnings.warn(f'capturing the RCCL collectives of task {task!r} failed ({type(e).__name__}: {e}); '
                          'falling back to graph(forward + backward) + eager exchange')
            torch.cuda.synchronize()
            self.exchange_in_graph, self.split = False, True
            self._warm_and_capture()
        self.graph.replay()  # capture only records: this replay is the iteration prepare_step() announced
        self._finish()
        self.done = torch.cuda.Event()
        self.done.record()
        self.warm_iters = 1  # iterations applied to the weights on this batch (the warm-ups were rolled back)
        self.first_out = self._output(batch)

    def _warm_and_capture(self):
        # warm-up (allocator, workspaces, lazy inits), then capture — both on the runner's stream, which is
        # the current stream here
        side = torch.cuda.current_stream()
        # the two warm-up iterations must not move the training trajectory (the reference applies ONE update per batch):
        # weights, moments and step counts are put back afterwards; only the first replay counts
        snap = self.opt.snapshot()
        ops.DEFER.pin = True  # the flush tables looked up from here on are baked into the graph by address
        ops.DEFER.prepare_capture(16)
        try:
            for _ in range(2):
                self.opt.prepare_step(self.table)
                self._body()
                self._finish()
                side.synchronize()  # the pinned optimizer table is refilled by the next prepare_step()
            torch.cuda.synchronize()
            if self.runner.sync is not None:
                # Let the process group's watchdog retire every collective issued so far before a stream goes into
                # capture: it polls the end events of its pending works every 100 ms, and HIP refuses an event query
                # ("operation not permitted on an event last recorded in a capturing stream") when the stream the event
                # was recorded on is capturing at that moment — which aborts the process from the watchdog thread.
                time.sleep(0.35)
            self.opt.restore(snap)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.prepare_step(self.table)
            # thread_local: the RCCL watchdog thread polls its events while this thread captures; under the default
            # global mode that poll is "not permitted when stream is capturing" and kills the process group
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local'):
                self._body()
        except Exception:
            ops.DEFER.drop()
            self.opt.restore(snap)
            raise
        finally:
            ops.DEFER.pin = False

    def _output(self, batch):
        # the packed loss vector is cloned (the static one is overwritten by the next replay) and read
        # lazily: the host does not wait for the graph, it goes on to queue the next iteration
        prefix = f"{self.task}.{batch.get('dataset_name')}"
        lv = LazyLogVars(self.names, self.packed.clone())
        if self.split:  # rank-averaged log variables (multitask_learner.py:299-304), one packed all-reduce
            lv = lv.all_reduced()
        return dict(loss=None, log_vars=lv.prefixed(prefix), num_samples=len(batch['img_metas']))

    def _draw(self):
        return self.model.cls_augments.draw(self.static['img'].shape[0], self.static['img'].shape[-2:])

    def _body(self):
        data = dict(self.meta, **self.static)
        data.pop('rnd', None)
        if self.aug is not None:
            data['rnd'] = dict(cls_aug_static=self.aug)
        if self.det_static is not None:
            data['static'] = self.det_static
        losses = self.model(**data)
        loss, self.names, packed = self.model.pack_losses(losses)
        self.packed = packed * self.weight
        self.opt.zero_grad()
        sync = self.runner.sync if self.exchange_in_graph else None
        if sync is not None:
            sync.begin_step(self.task)
        # weight-gradient contractions on a second stream, joined before anything reads the arena.  Off by
        # default: inside a hipGraph the forked branch brought nothing on ROCm 7.2 (71.9 ms/round without,
        # 72-73 with; profiles/README.md) — the replay does not overlap the two branches.
        side = os.environ.get('RSCOTR_SIDE_STREAM', '0') == '1'
        if side:
            ops.side_enable(True)
        try:
            (loss * self.weight).backward()
        finally:
            if side:
                ops.side_join()
                ops.side_enable(False)
        ops.flush_deferred()  # one combine launch for every split-K weight gradient of this backward pass
        log_work = None
        if sync is not None:
            sync.finish_step(self.task)  # leftover buckets in the fixed order, then the waits (event edges in the graph)
            import torch.distributed as dist
            # rank-averaged log variables (multitask_learner.py:299-304): same place in the collective sequence as on every
            # other path (after the buckets), but nothing on the compute queue depends on it — clip + AdamW are queued
            # first and the wait comes last, so the queue hand-over to RCCL and back is off the critical path
            self.packed = self.packed / dist.get_world_size()
            from .dist import INLINE
            if INLINE:
                dist.all_reduce(self.packed)  # (on the compute stream: the captured iteration stays one chain)
            else:
                log_work = dist.all_reduce(self.packed, async_op=True)
        if not self.split:
            self.opt.launch_step(self.table)
        if log_work is not None:
            log_work.wait()

    def _finish(self):
        if self.split:
            self.runner.sync.reduce_task(self.task)
            self.opt.launch_step(self.table)

    def accepts(self, batch):
        """det: the batch must fit the capacities this iteration was captured with."""
        if self.det_static is None:
            return True
        counts = [int(l.shape[0]) for l in batch['gt_labels']]
        gen = self.model.bbox_head.dn_generator
        mx = max(counts + [0])
        return mx <= self.det_static.gcap and 2 * gen.get_num_groups(mx) * mx <= self.det_static.padcap \
            and tuple(tuple(m['img_shape'][:2]) for m in batch['img_metas']) == tuple(self.det_static.img_shapes)

    def run(self, batch):
        if self.done is not None:
            self.done.synchronize()  # this graph's previous replay has consumed its pinned host buffers
        if self.det_static is not None:
            from .det_head import DetStatic
            DetStatic(self.model.bbox_head, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'],
                      batch['img'].device, gcap=self.det_static.gcap, padcap=self.det_static.padcap,
                      pinned=self.det_pinned).update_into(self.det_static)
        for k, t in self.static.items():
            t.copy_(batch[k], non_blocking=True)
        if self.aug is not None:
            sp = self.model.cls_augments.static_params(self._draw(), self.static['img'].shape[0])
            for k, t in self.aug.items():
                self.aug_host[k].copy_(sp[k])  # pinned staging: the upload below must not stall the host
                t.copy_(self.aug_host[k], non_blocking=True)
        self.opt.prepare_step(self.table)
        if _HOST_TIMES:
            t0 = time.perf_counter()
        self.graph.replay()
        if _HOST_TIMES:  # (diagnostic: host time of the graph launch alone)
            print(f'[runner] {self.task}: graph launch {1e3 * (time.perf_counter() - t0):.2f} ms on the host', flush=True)
        self._finish()
        self.done = torch.cuda.Event()
        self.done.record()
        return self._output(batch)


class IterBasedRunner:
    def __init__(self, model, optimizer, data_loader, lr_config=None, log_interval=0, logger=print,
                 bucket_mb=None, rnd_fn=None, graph_tasks=None):
        self.model, self.optimizer, self.data_loader = model, optimizer, data_loader
        self.iter = 0
        self.lr_updater = None
        if lr_config and lr_config.get('policy') == 'step':
            self.lr_updater = StepLrUpdater(**{k: v for k, v in lr_config.items() if k != 'policy'})
        self.log_interval, self.logger = log_interval, logger
        if bucket_mb is None:  # gradient exchange granularity (MB of fp32 gradients per all-reduce)
            from .dist import INLINE
            # inline exchange: few large collectives (nothing overlaps them anyway, and a flush of the deferred work
            # precedes every bucket); overlapped exchange: 32 MB so that the first buckets leave early in backward
            bucket_mb = float(os.environ.get('RSCOTR_BUCKET_MB', 128.0 if INLINE else 32.0))
        self.sync = GradSync(optimizer, bucket_mb) if is_dist() else None
        self.rnd_fn = rnd_fn
        # tasks whose iteration is replayed from a hipGraph (RSCOTR_GRAPHS=0 disables)
        if graph_tasks is None:
            graph_tasks = ('cls', 'det', 'seg') if os.environ.get('RSCOTR_GRAPHS', '1') != '0' else ()
            if os.environ.get('RSCOTR_GRAPH_TASKS') is not None:  # e.g. "cls,seg"
                graph_tasks = tuple(t for t in os.environ['RSCOTR_GRAPH_TASKS'].split(',') if t)
        self.graph_tasks = () if rnd_fn is not None else tuple(graph_tasks)
        self.graphed = {}
        self._seen = {}
        self.last_task = None
        self.force_eager = False  # bench.py: profiled eager rounds (per-kernel HIP events cannot ride in a graph)
        # The whole loop — eager iterations, graph warm-ups, captures and replays — runs on ONE side stream:
        # autograd binds every AccumulateGrad node to the stream it was created on, and a capture that has
        # to synchronise with a different (non-capturing) stream is invalid.
        self.stream = torch.cuda.Stream() if (self.graph_tasks and torch.cuda.is_available()) else None
        self._it = None
        self.log_buffer = OrderedDict()
        # hook surface of mmcv's runner that rscotr_amd.engine.MultiDatasetsEvalHook uses
        self.hooks, self.epoch, self.meta, self.work_dir = [], 0, {}, None
        self.log_buffer_output, self.log_buffer_ready = OrderedDict(), False

    def register_hook(self, hook):
        self.hooks.append(hook)
        if hasattr(hook, 'before_run'):
            hook.before_run(self)

    def train_iter(self):
        for h in self.hooks:
            if hasattr(h, 'before_train_iter'):
                h.before_train_iter(self)
        if self.stream is None or torch.cuda.current_stream() == self.stream:
            out = self._train_iter()  # (inside on_stream(): no stream hand-over per iteration)
        else:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                out = self._train_iter()
            torch.cuda.current_stream().wait_stream(self.stream)
        for h in self.hooks:
            if hasattr(h, 'after_train_iter'):
                h.after_train_iter(self)
        return out

    def _train_iter(self):
        if self._it is None:
            self._it = iter(self.data_loader)
        batch = next(self._it)
        if self.rnd_fn is not None:
            batch = dict(batch, rnd=self.rnd_fn(batch))
        if self.lr_updater is not None:
            self.optimizer.set_lr_factor(self.lr_updater.factor(self.iter))
        task = self.last_task = batch['task']
        if task in self.graph_tasks and batch['img'].is_cuda and not self.force_eager:
            # first iteration of a task runs eagerly (parameter liveness, workspaces); the second
            # captures (and applies 3 iterations' worth of updates on this batch); then replay
            self._seen[task] = self._seen.get(task, 0) + 1
            g = self.graphed.get(task)
            if g is None and self._seen[task] == 2:
                g = self.graphed[task] = GraphedTask(self, task, batch)
                self.iter += g.warm_iters
                out, g.first_out = g.first_out, None
                self.log_buffer = out['log_vars']
                return out
            if g is not None and g.accepts(batch):
                out = g.run(batch)
                self.iter += 1
                self.log_buffer = out['log_vars']
                return out
        if task == 'det' and 'det_trunk' in self.graph_tasks and batch['img'].is_cuda:
            self._seen[task] = self._seen.get(task, 0) + 1
            if self._seen[task] == 2 and getattr(self.model, '_trunk_graph', None) is None:
                keep = self.model._drop_keep(batch['img'].shape[0], batch['img'].device, None)
                if keep is not None:
                    self.model.enable_graphed_trunk(batch['img'], keep)
        # Distributed: every path of a task must issue the SAME collective sequence on every rank, whichever path each rank
        # takes for this batch (a rank whose det batch exceeds the captured capacities runs eagerly while the others replay
        # their graph): [det: normalisers, in forward] -> gradient buckets in arena order after backward -> packed log
        # vector (n floats).  Tasks that are never graphed keep the overlapped exchange (buckets launched from backward
        # hooks, back to front) — there all ranks run eagerly, always.
        graphed_task = self.sync is not None and task in self.graph_tasks and not self.force_eager
        if self.sync is not None:
            self.model.defer_log_allreduce = True  # the packed log vector is exchanged here, after the buckets
        out = self.model.train_step(batch, self.optimizer)
        # OptimizerHook.after_train_iter
        self.optimizer.zero_grad()
        if self.sync is not None and not (graphed_task and task in self.sync.plans):
            self.sync.begin_step(task)
        out['loss'].backward()
        ops.flush_deferred()
        if self.sync is not None:
            if graphed_task and task in self.sync.plans:
                self.sync.reduce_task(task)
            else:
                self.sync.finish_step(task)
        self.optimizer.step()
        if self.sync is not None and isinstance(out['log_vars'], LazyLogVars):
            out['log_vars'] = out['log_vars'].all_reduced()
        out['loss'] = out['loss'].detach()  # drop the autograd graph (and its AccumulateGrad nodes) now
        self.iter += 1
        self.log_buffer = out['log_vars']
        if self.log_interval and self.iter % self.log_interval == 0:
            self.logger(f'iter {self.iter} ' + ' '.join(f'{k}={v:.4f}' for k, v in out['log_vars'].items()
                                                       if k.endswith('.loss')))
        return out

    @contextlib.contextmanager
    def on_stream(self):
        """Make the runner's stream the current stream for a whole loop.  A train_iter() called from another stream hands
        over to the runner's stream and back with two cross-stream event waits; on this runtime the two streams sit on
        different hardware queues and every such hand-over stalls the device for ~0.3 ms (measured: 2.1 ms per round of
        three iterations, the gain of GPU_MAX_HW_QUEUES=1) — inside this context there is one hand-over per loop."""
        if self.stream is None or torch.cuda.current_stream() == self.stream:
            yield self
            return
        outer = torch.cuda.current_stream()
        self.stream.wait_stream(outer)
        try:
            with torch.cuda.stream(self.stream):
                yield self
        finally:
            outer.wait_stream(self.stream)

    def run(self, max_iters):
        with self.on_stream():
            while self.iter < max_iters:
                self.train_iter()


def build_runner(model, cfg, data_loader, **kwargs):
    optimizer = build_optimizer(model, cfg['optimizer'], cfg.get('optimizer_config'))
    return IterBasedRunner(model, optimizer, data_loader, lr_config=cfg.get('lr_config'), **kwargs)
