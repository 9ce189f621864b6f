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


def _gt_host(batch):
    hb, hl = batch.get('gt_bboxes_host'), batch.get('gt_labels_host')
    return None if hb is None or hl is None else (hb, hl)


def _recover_from_failed_capture(graph, cause=None):
    """After an exception inside a stream capture: make sure the stream has left capture mode and swallow the error the
    runtime still holds (HIP reports a failed call of a capturing stream once more at the NEXT API call).  A capture that
    cannot be ended — this runtime keeps the stream capturing after `hipErrorStreamCaptureUnjoined`, seen with gloo
    collectives on device tensors (tests/test_dist_gpu.py) — is fatal: nothing can be launched on that stream any more, so
    the run stops HERE with the cause instead of failing somewhere later or hanging the other ranks in a collective."""
    for _ in range(4):
        try:
            if graph is not None and torch.cuda.is_current_stream_capturing():
                graph.capture_end()
            torch.cuda.synchronize()
            return
        except RuntimeError:
            continue
    still = True
    try:
        still = torch.cuda.is_current_stream_capturing()
    except RuntimeError:
        pass
    if still:
        raise RuntimeError('capturing the iteration failed and the stream cannot leave capture mode; rerun with '
                           f'RSCOTR_DIST_CAPTURE=0 (graph = forward + backward, eager exchange).  Cause: {cause}') from cause


def _sync_works_only(sync):
    from .dist import INLINE
    return INLINE or (sync is not None and sync.comm is not None)


def _wait_watchdog_idle(limit=5.0, sync_only=None):
    """Block until the RCCL process group's watchdog has RETIRED every collective issued so far (call after a device
    synchronise, before a stream goes into capture).  The watchdog polls the end events of its pending works, and HIP refuses
    an event query ("operation not permitted on an event last recorded in a capturing stream") while the stream the event
    was recorded on is capturing — which aborts the process from the watchdog thread.  The works are complete after the
    synchronise, but the watchdog drops them only on its next pass; c10d's flight recorder marks an entry retired in that
    same pass, so "no active entries" = "the watchdog holds no event of ours".  -> ('recorder', seconds waited).  Where the
    recorder is off (TORCH_NCCL_TRACE_BUFFER_SIZE=0) or its dump is not readable: the inline exchange (synchronous works only)
    waits three watchdog periods (3 x 100 ms + margin) -> ('sleep', 0.35); the overlapped exchange refuses to capture."""
    import pickle
    dump = getattr(torch._C._distributed_c10d, '_dump_nccl_trace', None)
    t0 = time.perf_counter()
    timed_out = False
    try:
        if dump is not None:
            full = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=False))
            if full and full.get('entries'):  # the recorder is on: it has seen the warm-up iterations' collectives
                while time.perf_counter() - t0 < limit:
                    act = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=True))
                    if not (act and act.get('entries')):
                        return 'recorder', time.perf_counter() - t0
                    time.sleep(0.01)
                timed_out = True  # the recorder works, the watchdog just has not retired its entries in time
    except (RuntimeError, pickle.UnpicklingError, KeyError, TypeError, AttributeError) as e:  # a diagnostic API of c10d
        cause = e
    else:
        cause = None
    from .dist import INLINE
    if not (INLINE if sync_only is None else sync_only):
        # the overlapped exchange THROUGH c10d leaves asynchronous works with the watchdog: guessing when it has dropped their events is
        # not good enough (a wrong guess aborts the process from the watchdog thread in the middle of a capture)
        if timed_out:
            raise RuntimeError(f'capturing the overlapped gradient exchange through c10d: the RCCL watchdog still holds active works '
                               f'{limit:.0f} s after a device synchronise (flight recorder on) — not safe to start a capture; use '
                               'the exchange\'s own communicator (RSCOTR_DIST_DIRECT=1, the default) or the inline exchange')
        raise RuntimeError('capturing the overlapped gradient exchange through c10d (RSCOTR_DIST_INLINE=0, RSCOTR_DIST_DIRECT=0) needs '
                           'c10d\'s flight recorder to tell when the RCCL watchdog is idle, and it is not available '
                           '(TORCH_NCCL_TRACE_BUFFER_SIZE=0?'
                           + (f' {type(cause).__name__}: {cause}' if cause is not None else ' no entries recorded')
                           + '); enable it, or use the exchange\'s own communicator / the inline exchange')
    time.sleep(0.35)  # (inline exchange: synchronous works only — three watchdog periods are ample)
    return 'sleep', 0.35


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
    capture_veto = False  # injection point of the capture-fallback test: never set by product code or the environment

    def __init__(self, runner, task, batch):
        self.runner, self.task = runner, task
        self.model, self.opt = runner.model, runner.optimizer
        self.static = {k: batch[k].clone() for k in self.TENSOR_KEYS if k in batch}
        self.meta = {k: v for k, v in batch.items() if k not in self.static and not k.endswith('_host')}  # (host copies / host-reduced normalisers: per batch)
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
                from .dist import control_all_reduce
                control_all_reduce(need, dist.ReduceOp.MAX)
                gcap = _round_up(max(int(need.item()), 32), 32)
            self.det_static = DetStatic(self.model.bbox_head, batch['gt_bboxes'], batch['gt_labels'], batch['img_metas'],
                                        batch['img'].device, gcap=gcap, gt_host=_gt_host(batch))
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
        if self.exchange_in_graph:
            import torch.distributed as dist
            # only RCCL collectives can ride in a hipGraph (gloo stages through the host on streams of its own: the capture
            # ends "unjoined" and the stream stays in capture mode): any other backend takes the split form from the start
            self.exchange_in_graph = dist.get_backend() == 'nccl'
        self.split = runner.sync is not None and not self.exchange_in_graph
        self._capture_agreed()
        self._replay()  # capture only records: this replay is the iteration prepare_step() announced
        self.done = torch.cuda.Event()
        self.done.record()
        self.warm_iters = 1  # iterations applied to the weights on this batch (the warm-ups were rolled back)
        self.first_out = self._output(batch)

    def _capture_agreed(self):
        """Warm up and capture; with more than one rank the FORM of the captured iteration is one decision of all ranks.
        A capture of the RCCL collectives that fails on one rank only must not leave that rank re-running its warm-ups in
        the split form (two more iterations of bucket and log all-reduces) while the others go straight to their replay:
        the collective sequences would diverge and the job would hang.  Every rank therefore reports success (1) or the
        failure (0) of its attempt, MIN over ranks decides, and on 0 EVERY rank — also those whose capture succeeded —
        drops its graph, resets the exchange state and captures again in the split form.  Only errors of the capture
        itself are caught (RuntimeError from HIP / RCCL / torch's graph machinery); anything else propagates."""
        err = None
        # state of the optimizer BEFORE any attempt: a rank whose capture succeeded has already announced the captured
        # iteration (prepare_step after the roll-back of its warm-ups: step counts + 1); if the ranks then agree to fall back,
        # every rank starts its second attempt from this same state, or the Adam bias corrections would differ between the
        # ranks for the rest of the run
        snap0 = self.opt.snapshot() if self.exchange_in_graph else None
        try:
            self._warm_and_capture()
        except RuntimeError as e:
            if not self.exchange_in_graph:
                raise
            err = e
            _recover_from_failed_capture(getattr(self, 'graph', None), e)
        if not self.exchange_in_graph:
            ops.DEFER.keep_captured()
            return
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.static['img'].device)
        import torch.distributed as dist
        torch.cuda.synchronize()
        from .dist import control_all_reduce
        control_all_reduce(ok, dist.ReduceOp.MIN)
        if GraphedTask.capture_veto:  # (tests only — tests/test_dist_gpu.py sets the attribute: "another rank's capture failed")
            ok.zero_()
        if int(ok.item()) == 1:
            ops.DEFER.keep_captured()
            return
        import warnings
        warnings.warn(f'capturing the RCCL collectives of task {self.task!r} failed on at least one rank'
                      + (f' (here: {type(err).__name__}: {err})' if err is not None else ' (not here)')
                      + '; every rank falls back to graph(forward + backward) + eager exchange')
        torch.cuda.synchronize()
        self.graph = None
        self.runner.sync.reset_step()
        ops.DEFER.drop()
        ops.DEFER.forget_captured()  # (tables the dropped graph's copy nodes were to fill)
        self.exchange_in_graph, self.split = False, True
        self.opt.restore(snap0)
        self._warm_and_capture()
        steps = torch.tensor([float(self.opt.steps.sum())], dtype=torch.float64, device=self.static['img'].device)
        lo, hi = steps.clone(), steps.clone()
        control_all_reduce(lo, dist.ReduceOp.MIN)
        control_all_reduce(hi, dist.ReduceOp.MAX)
        if float(lo.item()) != float(hi.item()):
            raise RuntimeError(f'optimizer step counts differ between the ranks after the capture fallback ({lo.item()} / {hi.item()})')

    def _replay(self):
        self.graph.replay()
        if not self.split:
            # the replayed graph holds the optimizer step: the parameters have changed, and nothing on the host has said
            # so (WPLANES.bump() in launch_step ran at capture time only).  Without this, a non-captured forward after
            # replays — the evaluation hook, an eager det iteration whose batch exceeds the captured capacities — would find
            # its weight planes "fresh" and multiply with weights at least one step old
            ops.WPLANES.bump(by_optimizer=True)  # (the replayed update kernel rewrote the parameters' range words itself)
        self._finish()

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
                # c10d's watchdog must have retired the warm-ups' works (the det normalisers, plan checks: synchronous
                # collectives whose end events sit on THIS stream) before the stream starts capturing: it may not query them
                # afterwards.  With the overlapped exchange on its own communicator (dist.DirectComm) these are the only
                # c10d works there are — as in the inline form.
                self.watchdog_wait = _wait_watchdog_idle(sync_only=_sync_works_only(self.runner.sync))
            self.opt.restore(snap)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.prepare_step(self.table)
            # thread_local: the RCCL watchdog thread polls its events while this thread captures; under the default
            # global mode that poll is "not permitted when stream is capturing" and kills the process group
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local'):
                self._body()
        except Exception:
            ops.DEFER.drop()
            ops.DEFER.forget_captured()
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
        log_work, log_direct = None, False
        if sync is not None:
            sync.finish_step(self.task)  # leftover buckets in the fixed order, then the waits (event edges in the graph)
            import torch.distributed as dist
            # rank-averaged log variables (multitask_learner.py:299-304): same place in the collective sequence as on every
            # other path (after the buckets), but nothing on the compute queue depends on it — clip + AdamW are queued
            # first and the wait comes last, so the queue hand-over to RCCL and back is off the critical path
            from .dist import INLINE
            if INLINE:
                self.packed = self.packed / dist.get_world_size()
                dist.all_reduce(self.packed)  # (on the compute stream: the captured iteration stays one chain)
            elif sync.comm is not None:
                self.packed = self.packed.contiguous()
                sync.allreduce_avg_async(self.packed)  # (the exchange's own communicator: no c10d work inside the capture)
                log_direct = True
            else:
                self.packed = self.packed / dist.get_world_size()
                log_work = dist.all_reduce(self.packed, async_op=True)
        if not self.split:
            self.opt.launch_step(self.table)
        if log_work is not None:
            log_work.wait()
        if log_direct:
            sync.wait()

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
                      pinned=self.det_pinned, gt_host=_gt_host(batch),
                      norms_r_host=batch.get('det_norms_r_host')).update_into(self.det_static)
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
        self._replay()
        if _HOST_TIMES:  # (diagnostic: host time of the graph launch (+ the split form's eager tail))
            print(f'[runner] {self.task}: graph launch {1e3 * (time.perf_counter() - t0):.2f} ms on the host', flush=True)
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
        # host-side control group (gloo): batch-dependent decisions that every rank must take alike (replay the det graph or
        # run this batch eagerly) are agreed on WITHOUT touching the device queue — the data they depend on (ground-truth
        # counts, image shapes) is on the host already
        self.ctrl = None
        if self.sync is not None:
            import torch.distributed as dist
            # (also for the one-rank group of RSCOTR_DIST_SINGLE=1: the control traffic must take the same road as on N ranks)
            self.ctrl = dist.new_group(backend='gloo')
            ops.set_host_group(self.ctrl)  # (the det normalisers are averaged through it too: rscotr_amd.det_head.DetStatic)
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
        self.outputs, self.max_iters, self.timestamp = None, None, None

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
        self.outputs = out
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
            g = self._graph_for(task, batch)
            if g is not None:
                out = g.run(batch)
                self.iter += 1
                self.log_buffer = out['log_vars']
                return out
        # Distributed: the ranks take the same path for every batch (_graph_for agrees on it), and the eager path of a graphed
        # task issues the collective sequence of its graph anyway: [det: normalisers, in forward] -> gradient buckets in
        # arena order after backward -> packed log vector (n floats).  Tasks that are never graphed keep the overlapped
        # exchange (buckets launched from backward hooks, back to front).
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

    def _agree(self, ok):
        """MIN over ranks of a local yes / no (one tiny gloo all-reduce on the host; identity in a single process)."""
        if self.ctrl is None:
            return bool(ok)
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.ctrl)
        return bool(int(t[0]))

    def _graph_for(self, task, batch):
        """-> the GraphedTask that replays this batch, or None (run it eagerly): the SAME answer on every rank.  Only the det
        graph's answer depends on the batch (it must fit the captured ground-truth / denoising capacities); a rank whose
        batch does not fit pulls every rank to the eager path for this iteration, so graph and eager iterations never meet
        in one collective sequence."""
        g = self.graphed.get(task)
        if g is None:
            return None
        ok = g.accepts(batch)
        if g.det_static is not None:
            if self.ctrl is not None:
                # ONE host message per det iteration carries both things the ranks must share before it starts: the four loss
                # normalisers (reduce_mean of detr_head.py:379-390 / dino_head.py:266-283 — functions of the ground-truth counts,
                # host data) and this rank's "my batch does not fit the captured capacities" flag.  Nothing of it touches the
                # device queue: the iteration's device work is the one-rank iteration's plus the gradient exchange
                from .det_head import DetStatic
                counts = [int(l.shape[0]) for l in batch['gt_labels']]
                r = DetStatic.reduce_norms_host(DetStatic.host_norms(self.model.bbox_head, counts), extra=[0.0 if ok else 1.0])
                batch['det_norms_r_host'] = r[:4].copy()
                ok = float(r[4]) == 0.0
            else:
                ok = self._agree(ok)
        return g if ok else None

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

    def run(self, max_iters=None):
        """Train until `max_iters` iterations are done (default: the `runner.max_iters` of the config)."""
        if max_iters is None:
            max_iters = self.max_iters
        assert max_iters is not None, 'no iteration count: pass max_iters or build the runner from a config with `runner`'
        if self.max_iters is None:
            self.max_iters = max_iters
        with self.on_stream():
            while self.iter < max_iters:
                self.train_iter()
        for h in self.hooks:
            if hasattr(h, 'after_run'):
                h.after_run(self)

    # ---- mmcv BaseRunner.load_checkpoint / resume (mtl/apis/train.py:115-118) --------------------------------------
    def load_checkpoint(self, path, map_location='cpu', strict=False):
        from .checkpoint import load_checkpoint
        self.logger(f'load checkpoint from {path}')
        return load_checkpoint(self.model, path, strict=strict, map_location=map_location)[0]

    def resume(self, path, map_location='cpu'):
        from .checkpoint import resume
        ckpt = resume(self, path, map_location=map_location)
        self.epoch = int(ckpt.get('meta', {}).get('epoch', 0))
        if 'hook_msgs' in ckpt.get('meta', {}):
            self.meta = dict(self.meta or {}, hook_msgs=ckpt['meta']['hook_msgs'])
        # graphs captured before the resume replay the OLD step counts' bias corrections only through the per-iteration
        # table (prepare_step), so they stay valid; the captured det capacities do too
        for h in self.hooks:
            if hasattr(h, 'after_resume'):
                h.after_resume(self)
        self.logger(f'resumed from {path}: iter {self.iter}')
        return ckpt


def build_runner(model, cfg, data_loader, val_dataloaders=None, validate=None, work_dir=None, meta=None, timestamp=None,
                 **kwargs):
    """What `mtl/apis/train.py:24-118::train_model` does between building the model and `runner.run`: optimizer
    (`build_optimizer(model, cfg.optimizer)`), runner (`cfg.runner`), `register_training_hooks(lr_config, optimizer_config,
    checkpoint_config, log_config)` (:77-83), the evaluation hook from `cfg.evaluation` when validating (:89-107), then
    `auto_resume` / `resume_from` / `load_from` (:109-118).  `val_dataloaders`: {dataset name: loader} for the evaluation
    hook (`validate` defaults to "loaders were given"); `work_dir` defaults to `cfg.work_dir`.  LR schedule and optimizer
    hook are built into the runner (`lr_config`, `optimizer_config`); checkpoint / logger hooks come from
    rscotr_amd.hooks under the reference's type strings."""
    from .hooks import CheckpointHook, build_hook, find_latest_checkpoint
    get = cfg.get if hasattr(cfg, 'get') else (lambda k, d=None: getattr(cfg, k, d))
    optimizer = build_optimizer(model, cfg['optimizer'], get('optimizer_config'))
    log_cfg = dict(get('log_config') or {})
    runner = IterBasedRunner(model, optimizer, data_loader, lr_config=get('lr_config'), **kwargs)
    rcfg = dict(get('runner') or {})
    if rcfg.get('type', 'IterBasedRunner') != 'IterBasedRunner':
        raise NotImplementedError(f"runner type {rcfg.get('type')!r}: the MTL configs train with IterBasedRunner")
    runner.max_iters = rcfg.get('max_iters')
    runner.work_dir = work_dir if work_dir is not None else get('work_dir')
    runner.meta = dict(meta or {})
    runner.timestamp = timestamp
    # register_training_hooks: checkpoint (priority NORMAL), then the logger hooks (VERY_LOW); the evaluation hook (LOW)
    # runs between them in mmcv's priority order — after the checkpoint hook, before the loggers, which then report its
    # metrics in the same iteration
    ck = get('checkpoint_config')
    if ck is not None:
        ck = dict(ck)
        ck.setdefault('by_epoch', False)  # mmcv IterBasedRunner.register_training_hooks
        runner.register_hook(CheckpointHook(**{k: v for k, v in ck.items() if k != 'type'}))
    validate = (val_dataloaders is not None) if validate is None else validate
    if validate:
        if val_dataloaders is None:
            raise ValueError('validate=True needs val_dataloaders ({dataset name: loader})')
        from .engine import MultiDatasetsEvalHook
        eval_cfg = dict(get('evaluation') or {})
        eval_cfg['by_epoch'] = False  # `eval_cfg['by_epoch'] = cfg.runner['type'] != 'IterBasedRunner'` (:97)
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise NotImplementedError('distributed validation (mtl/apis/train.py:98-99 raises the same)')
        runner.register_hook(MultiDatasetsEvalHook(val_dataloaders, **eval_cfg))
    interval = log_cfg.get('interval', 10)
    for h in log_cfg.get('hooks', [dict(type='TextLoggerHook')] if log_cfg else []):
        runner.register_hook(build_hook(h, interval=interval, by_epoch=False))
    # resume / load (train.py:109-118)
    resume_from = get('resume_from')
    if resume_from is None and get('auto_resume'):
        resume_from = find_latest_checkpoint(runner.work_dir)
    if resume_from:
        runner.resume(resume_from)
    elif get('load_from'):
        runner.load_checkpoint(get('load_from'))
    return runner
