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


class DirectComm:
    """An RCCL communicator of this job's ranks driven through the C ABI (rscotr_comm_*, csrc/comm.cpp) with a stream of its
    own: the OVERLAPPED exchange issues its collectives here, ordered against the compute stream by plain events (fork after
    a bucket's last gradient, join before clip + AdamW) — also inside a hipGraph capture.  No c10d work objects exist for
    these collectives, so c10d's watchdog thread has no event of a capturing stream to poll (that poll aborted 2 of 8
    overlapped runs in round 4).  The unique id travels over the job's existing process group."""

    def __init__(self, device):
        import ctypes
        from ._lib import lib
        self.lib = lib
        idt = torch.zeros(128, dtype=torch.uint8)
        if dist.get_rank() == 0:
            lib.call('rscotr_comm_unique_id', idt.data_ptr())
        d = idt.to(device)
        dist.broadcast(d, 0)
        idt = d.cpu()
        handle = ctypes.c_void_p()
        lib.call('rscotr_comm_init', idt.data_ptr(), dist.get_rank(), dist.get_world_size(), ctypes.byref(handle))
        self.handle = handle.value
        self.stream = torch.cuda.Stream(device)
        self.forked = False

    def allreduce_avg(self, t):
        """t (contiguous fp32, complete on the CURRENT stream) = mean over the ranks, in place, on the communicator's stream."""
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        self.lib.call('rscotr_comm_allreduce_avg', self.handle, t.data_ptr(), t.numel(), self.stream.cuda_stream)
        self.forked = True

    def join(self):
        """The current stream waits for everything issued on the communicator's stream."""
        if self.forked:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            torch.cuda.current_stream().wait_event(ev)
            self.forked = False

    def close(self):
        if self.handle:
            self.lib.call('rscotr_comm_destroy', self.handle)
            self.handle = 0


_ACTIVE_COMM = [None]  # the DirectComm of the overlapped exchange, when there is one (GradSync creates it)


def direct_comm():
    return _ACTIVE_COMM[0]


def control_all_reduce(t, op):
    """All-reduce of a small CONTROL value (capacities, success flags, plan hashes, step counts — start-up agreements, a few
    per run).  Through the host control group (gloo) when the runner made one: the device tensor is staged through the
    host, and the RCCL process group never sees these — with the overlapped exchange on its own communicator c10d's RCCL
    watchdog then has no work at all whose end event could sit on a stream that goes into capture."""
    from .ops.distutil import host_group
    hg = host_group()
    if hg is None or not t.is_cuda:
        dist.all_reduce(t, op=op)
        return t
    c = t.detach().cpu()
    dist.all_reduce(c, op=op, group=hg)
    t.copy_(c)
    return t


def mean_over_ranks(t):
    """t (small contiguous fp32 device vector) = its mean over the ranks, in place: on the exchange's own communicator when
    there is one (ordered after everything on the current stream, joined before returning), else one c10d all-reduce."""
    comm = direct_comm()
    if comm is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        comm.allreduce_avg(t)
        comm.join()
        return t
    t.div_(dist.get_world_size())
    dist.all_reduce(t)
    return t


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
        # the overlapped exchange runs on a communicator of its own (DirectComm); the inline one stays with c10d
        self.comm = None
        if (not INLINE and is_dist() and dist.get_backend() == 'nccl' and optimizer.flat_g.is_cuda
                and os.environ.get('RSCOTR_DIST_DIRECT', '1') != '0'):
            from ._lib import lib
            if lib.rscotr_comm_available():
                self.comm = _ACTIVE_COMM[0] = DirectComm(optimizer.flat_g.device)

    def allreduce_avg_async(self, t):
        """Overlapped form: mean of a small vector (the packed log variables) over the ranks, in the exchange's own sequence
        (after the buckets); `wait()` joins."""
        self.comm.allreduce_avg(t)

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.comm is not None:
            self.comm.join()

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
        control_all_reduce(t, dist.ReduceOp.MAX)
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
        self.wait()

    def _launch(self, b):
        if not is_dist():
            return
        view = self.opt.flat_g[b['lo']:b['hi']]
        if dist.get_backend() == 'nccl' and INLINE:
            # the collective on the COMPUTE stream, in launch order (c10d runs a synchronous collective on the current
            # stream): no fork, so a captured iteration stays ONE chain on one hardware queue
            dist.all_reduce(view, op=dist.ReduceOp.AVG, async_op=False)
        elif self.comm is not None:  # overlapped: ncclAllReduce(ncclAvg) on the communicator's stream, forked by an event
            self.comm.allreduce_avg(view)
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
        self.wait()
        self._bucket_of = self._vpending = None

    def describe(self):
        return {t: dict(buckets=len(p), mbytes=sum(b['hi'] - b['lo'] for b in p) * 4 / 2 ** 20)
                for t, p in self.plans.items()}
