"""Data-parallel gradient exchange for the co-training step (RCCL over xGMI via torch.distributed).

The reference wraps MTL in torch DDP (mtl/apis/train.py:37-46), which walks the autograd graph
each step to find the parameters the current task did not touch.  Here the parameter subset of
each task is static, so the plan is static too:
  * gradients live in one flat fp32 arena laid out task-major (rscotr_amd.optim.task_major_order);
  * the first step of each task runs un-overlapped and records which parameters received a
    gradient; from then on that task has a fixed list of buckets (contiguous arena slices of
    ~bucket_mb, cut at tensor boundaries, ordered back-to-front like backward produces them);
  * a post-accumulate hook counts parameters down per bucket and launches `all_reduce(AVG)` for a
    bucket as soon as its last gradient is written, so the exchange overlaps the rest of backward;
    buckets are always LAUNCHED in one fixed order (back to front of the arena, the order backward
    completes them in): a bucket that becomes ready early waits for its predecessors, so a rank-local
    difference in completion order can never pair mismatched collectives across ranks;
  * the plans are compared across ranks once per task (hash all-reduced MIN / MAX);
  * all ranks must draw the same task each iteration (same strategy state / NumPy seed on every
    rank, as the reference requires — tools/train.py:211-215).
"""
import os

import torch
import torch.distributed as dist


# RSCOTR_DIST_INLINE=1 (default): bucket all-reduces on the compute stream; =0: on RCCL's own stream, overlapping the rest of
# backward.  On this runtime a captured iteration with a forked RCCL branch is replayed across several hardware queues and the
# hand-overs cost more than the exchange they hide (one-rank group, per round: plain 40.6 ms, inline 41.7, overlapped 44.8;
# DESIGN.md section 6)
INLINE = os.environ.get('RSCOTR_DIST_INLINE', '1') == '1'


def is_dist():
    """True when gradients have to be exchanged.  RSCOTR_DIST_SINGLE=1 takes the distributed code path with a
    one-rank group (exercises bucket plans, RCCL calls and the split graph/optimizer flow on a 1-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('RSCOTR_DIST_SINGLE') == '1'


class GradSync:
    def __init__(self, optimizer, bucket_mb=32.0):
        self.opt = optimizer
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        self.plans = {}       # task -> list of buckets
        self.active = None    # buckets of the running step
        self.fired = None     # discovery: set of param indices that received a gradient
        self.handles = []
        self.param_index = {id(g['param']): i for i, g in enumerate(optimizer.groups)}
        self._bucket_of = None
        self.fires = {}       # task -> {param index: gradient-ready notifications per step}
        self._order, self._next, self._warned = [], 0, False
        self._at_end = False
        optimizer.ready_callbacks.append(self._on_ready)
        # Overlap needs the gradients to LAND while backward is still running, but the direct-write path defers them
        # (split-K combines, LayerNorm folds and the grouped weight gradients wait for ops.flush_deferred()).  The
        # discovery step also counts the grad_written() calls per parameter; from then on the deferred work is flushed
        # every time a bucket has seen all of its writes, which completes that bucket and launches its all-reduce.
        optimizer.written_callbacks.append(self._on_written)
        self.vfires, self.vfired, self._vpending = {}, None, None

    def _on_ready(self, i):
        """A gradient contribution of parameter i is complete (AccumulateGrad ran, or a backward
        kernel added it into the arena directly; a parameter used k times through the direct path
        notifies k times per step — the discovery step records k)."""
        if self.fired is not None:
            self.fired[i] = self.fired.get(i, 0) + 1
        elif self._bucket_of is not None:
            b = self._bucket_of.get(i)
            if b is not None:
                b['pending'] -= 1
                if b['pending'] <= 0:
                    self._launch_ready()

    def _launch_ready(self):
        """Launch, in the fixed order, every bucket up to the first one that is not complete yet."""
        while self._next < len(self._order) and self._order[self._next]['pending'] <= 0:
            self._launch(self._order[self._next])
            self._next += 1

    def _check_plan(self, task):
        """All ranks must have cut the same buckets (same parameters fired): one tiny all-reduce per task."""
        if not is_dist():
            return
        import zlib
        h = zlib.crc32(repr([(b['lo'], b['hi']) for b in self.plans[task]]).encode()) & 0x7FFFFFFF
        t = torch.tensor([h, -h], dtype=torch.int64, device=self.opt.flat_g.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lo, hi = -int(t[1]), int(t[0])
        if lo != hi:
            raise RuntimeError(f'gradient bucket plans of task {task!r} differ across ranks: the ranks did not run the '
                               'same task / parameter subset this iteration')

    def _on_written(self, i):
        if self.vfired is not None:
            self.vfired[i] = self.vfired.get(i, 0) + 1
        elif self._vpending is not None:
            b = self._bucket_of.get(i) if self._bucket_of is not None else None
            if b is not None:
                k = id(b)
                self._vpending[k] -= 1
                if self._vpending[k] == 0:
                    from . import ops
                    ops.flush_deferred()

    def reduce_task(self, task):
        """Exchange the task's gradient buckets now (no overlap with backward) and make the current stream
        wait for them: the hipGraph-replayed iterations call this between backward (in the graph) and the
        optimizer step."""
        self.handles = []
        for b in reversed(self.plans[task]):  # the one launch order of every path: back to front
            self._launch(b)
        for h in self.handles:
            h.wait()
        self.handles = []

    def _launch(self, b):
        if not is_dist():
            return
        view = self.opt.flat_g[b['lo']:b['hi']]
        if dist.get_backend() == 'nccl' and INLINE:
            # the collective on the COMPUTE stream, in launch order (c10d runs a synchronous collective on the current
            # stream): no fork, so a captured iteration stays ONE chain on one hardware queue
            dist.all_reduce(view, op=dist.ReduceOp.AVG, async_op=False)
        elif dist.get_backend() == 'nccl':  # RCCL: mean in the collective
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.AVG, async_op=True))
        else:  # gloo (CPU tests) has no AVG
            view.div_(dist.get_world_size())
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True))

    def _build_plan(self, fired):
        """Cut the fired parameters (sorted by arena offset) into contiguous buckets."""
        opt = self.opt
        idx = sorted(fired, key=lambda i: opt.offsets[i])
        buckets, cur = [], None
        for i in idx:
            lo = opt.offsets[i]
            hi = lo + (opt.groups[i]['param'].numel() + 3) // 4 * 4
            if cur is not None and cur['hi'] == lo and (hi - cur['lo']) <= self.bucket_elems:
                cur['hi'] = hi
                cur['params'].append(i)
            else:
                cur = dict(lo=lo, hi=hi, params=[i])
                buckets.append(cur)
        return buckets

    def begin_step(self, task):
        """Call before backward."""
        self.handles = []
        plan = self.plans.get(task)
        if plan is None:
            self.fired, self._bucket_of = {}, None
            self.vfired, self._vpending = {}, None
        elif INLINE and dist.is_initialized() and dist.get_backend() == 'nccl':
            # inline exchange: nothing overlaps the collectives, so nothing is gained by launching a bucket early — and the
            # bucket-aligned flushes would cut the grouped weight-gradient launch into pieces.  No hooks are armed: backward
            # runs as on one GPU, finish_step() sends the buckets in the fixed order after the one flush of the deferred work
            self.fired = self.vfired = None
            self._bucket_of = self._vpending = None
            self._order, self._next = list(reversed(plan)), 0
            self._at_end = True
        else:
            vf = self.vfires.get(task, {})
            self.vfired = None
            self._vpending = {id(b): sum(vf.get(i, 0) for i in b['params']) for b in plan}
            self.fired = None
            self._bucket_of = {}
            fires = self.fires[task]
            for b in plan:
                b['pending'] = sum(fires[i] for i in b['params'])
                for i in b['params']:
                    self._bucket_of[i] = b
            self._order, self._next = list(reversed(plan)), 0

    def reset_step(self):
        """Forget a step that was begun and never finished (a failed graph capture): no armed hooks, no pending handles."""
        self.handles = []
        self.fired = self.vfired = None
        self._bucket_of = self._vpending = None
        self._order, self._next = [], 0
        self._at_end = False

    def finish_step(self, task):
        """Call after backward, before the optimizer step: waits for the exchange."""
        if self.fired is not None:  # discovery step: plan from what fired, reduce everything now
            self.plans[task] = self._build_plan(self.fired)
            self.fires[task] = dict(self.fired)
            self.vfires[task] = dict(self.vfired or {})
            self.fired = self.vfired = None
            self._check_plan(task)
            for b in reversed(self.plans[task]):
                self._launch(b)
        elif self._at_end:
            self._at_end = False
            while self._next < len(self._order):
                self._launch(self._order[self._next])
                self._next += 1
        elif self._bucket_of is not None:
            # backward is over: whatever has not been launched (a parameter fired fewer times than in the discovery
            # step, so its bucket never counted down to zero) goes out now, in the same fixed order on every rank —
            # never skipped: a skipped bucket would leave that gradient un-averaged and the ranks would diverge
            late = [b for b in self._order[self._next:] if b['pending'] != 0]
            if late and not self._warned:
                self._warned = True
                import warnings
                warnings.warn(f'GradSync: {len(late)} bucket(s) of task {task!r} did not count down to zero during '
                              'backward (fire counts differ from the discovery step); exchanged after backward')
            while self._next < len(self._order):
                self._launch(self._order[self._next])
                self._next += 1
        for h in self.handles:
            h.wait()
        self.handles = []
        self._bucket_of = self._vpending = None

    def describe(self):
        return {t: dict(buckets=len(p), mbytes=sum(b['hi'] - b['lo'] for b in p) * 4 / 2 ** 20)
                for t, p in self.plans.items()}
