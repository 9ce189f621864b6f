"""N > 1 path on CPU: two gloo processes exercise the static per-task bucket plan of
rscotr_amd.dist.GradSync (discovery step, bucketed overlap step, unused parameters per task) and the
packed scalar all-reduce of MTL._parse_losses.  The arenas live on the CPU here; the optimizer's HIP
kernels are not called (they are covered by tests/test_optim_gpu.py)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class TwoTask(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Linear(8, 16)
        self.head_a = nn.Linear(16, 4)
        self.head_b = nn.Linear(16, 3)
        self.never = nn.Linear(2, 2)

    def forward(self, x, task):
        h = torch.relu(self.backbone(x))
        return (self.head_a if task == 'a' else self.head_b)(h).pow(2).sum()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from rscotr_amd.dist import GradSync
        from rscotr_amd.mtl import MTL
        from rscotr_amd.optim import FlatAdamW, build_param_groups
        torch.manual_seed(0)
        model = TwoTask()
        opt = FlatAdamW(build_param_groups(model, dict(type='AdamW', lr=1e-3, weight_decay=0.0)))
        sync = GradSync(opt, bucket_mb=0.0005)  # ~130 floats per bucket: several buckets per task
        ref = TwoTask()
        ref.load_state_dict(model.state_dict())
        for it, task in enumerate(['a', 'b', 'a', 'b', 'a']):
            xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(100 * it + r)) for r in range(world)]
            opt.zero_grad()
            sync.begin_step(task)
            model(xs[rank], task).backward()
            sync.finish_step(task)
            # reference: mean over ranks of the per-rank gradients, computed locally
            ref.zero_grad()
            for r in range(world):
                (ref(xs[r], task) / world).backward()
            for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
                want = torch.zeros_like(p) if pr.grad is None else pr.grad
                assert torch.allclose(p.grad, want, atol=1e-6), (it, task, n)
        # the hipGraph-replayed iterations exchange gradients after backward, outside the graph: reduce_task
        # (no begin/finish_step, no overlap) must give the same rank-mean on the task's static buckets
        for it, task in enumerate(['b', 'a']):
            xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(900 + 10 * it + r)) for r in range(world)]
            opt.zero_grad()
            model(xs[rank], task).backward()
            sync.reduce_task(task)
            ref.zero_grad()
            for r in range(world):
                (ref(xs[r], task) / world).backward()
            for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
                want = torch.zeros_like(p) if pr.grad is None else pr.grad
                assert torch.allclose(p.grad, want, atol=1e-6), ('reduce_task', task, n)
        # Robustness of the overlapped exchange: (1) a bucket that completes EARLY must not be launched before
        # its predecessors in the fixed back-to-front order — rank 0 sees the completions in reversed order here and the
        # collectives still pair up; (2) a bucket whose count never reaches zero (a parameter fired fewer times than in
        # the discovery step) is exchanged by finish_step instead of being skipped silently.
        import warnings
        order = []
        real_launch = sync._launch
        sync._launch = lambda b: (order.append(b['lo']), real_launch(b))[1]
        xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(4000 + r)) for r in range(world)]
        opt.zero_grad()
        g_local = torch.autograd.grad(model(xs[rank], 'a'), [p for n, p in model.named_parameters() if not n.startswith(('never', 'head_b'))])
        sync.begin_step('a')
        plan = sync.plans['a']
        assert len(plan) >= 2
        with torch.no_grad():  # write the local gradients into the arena, then notify in a rank-dependent order
            for p, g in zip([p for n, p in model.named_parameters() if not n.startswith(('never', 'head_b'))], g_local):
                p.grad.copy_(g)
        fire = [i for b in (plan if rank == 0 else list(reversed(plan))) for i in b['params']]
        skipped = plan[0]['params'][0] if rank == 1 else None  # rank 1 "forgets" one notification
        for i in fire:
            if i != skipped:
                for _ in range(sync.fires['a'][i]):
                    sync._on_ready(i)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            sync.finish_step('a')
        assert order == [b['lo'] for b in reversed(plan)], (rank, order)
        assert (rank == 1) == any('did not count down' in str(x.message) for x in w), (rank, [str(x.message) for x in w])
        sync._launch = real_launch
        ref.zero_grad()
        for r in range(world):
            (ref(xs[r], 'a') / world).backward()
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            want = torch.zeros_like(p) if pr.grad is None else pr.grad
            assert torch.allclose(p.grad, want, atol=1e-6), ('fixed order', n)
        # Bucket-aligned flush (the overlap of the direct-write path): gradients written straight into the arena are
        # announced through grad_written(); while deferred work is pending the notifications wait for ops.flush_deferred(),
        # and GradSync flushes as soon as a bucket has seen all the writes the discovery step counted for it — the
        # bucket's all-reduce is then launched from INSIDE backward, not after it.
        from rscotr_amd import ops as _ops
        names_i = {g['name']: i for i, g in enumerate(opt.groups)}
        task_params = [names_i[n] for n, _ in model.named_parameters() if not n.startswith(('never', 'head_a'))]
        sync.plans.pop('c', None)
        opt.zero_grad()
        sync.begin_step('c')  # discovery of a third "task" whose gradients all arrive through grad_written()
        for i in task_params:
            opt.grad_written(i)
        sync.finish_step('c')
        assert sync.vfires['c'] == {i: 1 for i in task_params}
        launched = []
        sync._launch = lambda b: (launched.append(b['lo']), real_launch(b))[1]
        opt.zero_grad()
        sync.begin_step('c')
        _ops.DEFER.ln_entries.append((0, 0, 0, 0, 0))  # something is pending: notifications are deferred
        flushes = []
        real_flush = _ops.flush_deferred

        def fake_flush():  # what ops.flush_deferred does for the notifications, without launching HIP kernels
            flushes.append(len(launched))
            _ops.DEFER.ln_entries.clear()
            pend, _ops.DEFER.notify = _ops.DEFER.notify, []
            for j in pend:
                opt._on_ready(j)
        _ops.flush_deferred = fake_flush
        try:
            last_bucket = sync.plans['c'][-1]
            for i in reversed(task_params):  # back to front, as backward produces them
                opt.grad_written(i)
                if i == last_bucket['params'][0]:
                    assert flushes and launched[:1] == [last_bucket['lo']], (flushes, launched)  # flushed + launched mid-way
                    _ops.DEFER.ln_entries.append((0, 0, 0, 0, 0))
        finally:
            _ops.flush_deferred = real_flush
            _ops.DEFER.ln_entries.clear()
        sync.finish_step('c')
        sync._launch = real_launch
        assert launched == [b['lo'] for b in reversed(sync.plans['c'])], launched
        # The INLINE exchange (RSCOTR_DIST_INLINE=1: buckets on the compute stream after backward, the default on RCCL) and the
        # OVERLAPPED one (buckets launched from backward hooks as they complete) must put the SAME collectives on the wire in the
        # SAME order — a job may mix the forms across iterations (eager / captured / fallback), never across ranks, and a
        # recorded sequence is what a rank-divergence would show up in.  Both forms are driven here on the RCCL branch of
        # GradSync (backend query patched, launches recorded instead of issued) with rank-dependent completion orders.
        import rscotr_amd.dist as D
        real_backend, real_inline = D.dist.get_backend, D.INLINE
        seqs = {}
        try:
            D.dist.get_backend = lambda *a, **k: 'nccl'
            for form, inline in (('inline', True), ('overlap', False)):
                D.INLINE = inline
                rec_launch = []
                sync._launch = lambda b: rec_launch.append((b['lo'], b['hi']))
                for task_ in ('a', 'b'):
                    plan_ = sync.plans[task_]
                    sync.begin_step(task_)
                    before_ = len(rec_launch)
                    fire_ = [i for b in (plan_ if rank == 0 else list(reversed(plan_))) for i in b['params']]
                    for i in fire_:
                        for _ in range(sync.fires[task_][i]):
                            sync._on_ready(i)
                    if inline:
                        assert len(rec_launch) == before_, 'the inline form launched a bucket during backward'
                    sync.finish_step(task_)
                seqs[form] = list(rec_launch)
        finally:
            D.dist.get_backend, D.INLINE = real_backend, real_inline
            sync._launch = real_launch
        assert seqs['inline'] == seqs['overlap'] and len(seqs['inline']) == len(sync.plans['a']) + len(sync.plans['b']), seqs
        both = [None, None]
        dist.all_gather_object(both, seqs['inline'])
        assert both[0] == both[1], 'the ranks would issue different collective sequences'
        # reduce_mean of a small device vector (det loss normalisers)
        from rscotr_amd import ops
        assert torch.allclose(ops.dist_mean_tensor(torch.tensor([2.0 * rank, 4.0])), torch.tensor([1.0, 4.0]))
        plans = sync.describe()
        assert set(plans) == {'a', 'b', 'c'} and plans['a']['buckets'] >= 2
        # a parameter no task touches never enters a plan; head_b is not in task a's plan
        names = {i: g['name'] for i, g in enumerate(opt.groups)}
        in_a = {names[i] for b in sync.plans['a'] for i in b['params']}
        assert not any(n.startswith(('never', 'head_b')) for n in in_a) and any(n.startswith('head_a') for n in in_a)
        # packed scalar reduction of the loss dict (one all-reduce, rank-consistency guard rides along)
        losses = {'loss_x': torch.tensor(float(rank + 1)), 'acc': torch.tensor([10.0 * (rank + 1)])}
        loss, logs = MTL._parse_losses(None, losses)
        assert abs(logs['loss_x'] - 1.5) < 1e-6 and abs(logs['acc'] - 15.0) < 1e-6 and abs(logs['loss'] - 1.5) < 1e-6
        assert float(loss) == float(rank + 1)  # the differentiable loss stays local
        # The graph-or-eager decision of a det iteration is ONE decision of all ranks (runner._graph_for): rank 1's batch
        # "exceeds the captured capacities" -> no rank replays its graph; both fit -> both replay.  The decision travels over
        # the host-side control group, so the collective sequences of the two paths never meet.
        from rscotr_amd.runner import IterBasedRunner
        runner = IterBasedRunner(model, opt, data_loader=None, graph_tasks=())
        assert runner.sync is not None and runner.ctrl is not None

        class FakeGraph:
            det_static = object()

            def __init__(self, fits):
                self.fits, self.asked = fits, 0

            def accepts(self, batch):
                self.asked += 1
                return self.fits
        # ... and the SAME host message carries the four det loss normalisers (reduce_mean of detr_head.py:379-390 /
        # dino_head.py:266-283): every rank leaves _graph_for with their rank average, and no device collective is needed for them
        import time
        import types
        import numpy as np
        from rscotr_amd.det_head import DetStatic
        head = types.SimpleNamespace(num_query=600, bg_cls_weight=0.0,
                                     dn_generator=types.SimpleNamespace(get_num_groups=lambda mx: max(1, 100 // max(mx, 1))))
        real_model, runner.model = runner.model, types.SimpleNamespace(bbox_head=head)
        counts = {0: [3, 5], 1: [7, 1]}
        want_norms = (DetStatic.host_norms(head, counts[0]) / np.float32(2) + DetStatic.host_norms(head, counts[1]) / np.float32(2))
        lat = []
        for fits_here, want in (((rank == 0), False), (True, True), (False, False)):
            runner.graphed['det'] = FakeGraph(fits_here)
            batch = dict(gt_labels=[torch.zeros(n, dtype=torch.long) for n in counts[rank]])
            t0 = time.perf_counter()
            got = runner._graph_for('det', batch)
            lat.append(time.perf_counter() - t0)
            assert (got is not None) == want and runner.graphed['det'].asked == 1, (rank, fits_here, want)
            assert np.array_equal(batch['det_norms_r_host'], want_norms), (batch['det_norms_r_host'], want_norms)
        if rank == 0:
            print(f'[test_dist_cpu] graph-or-eager + normaliser message, world 2 over gloo on this host: {min(lat) * 1e6:.0f} us')
        runner.model = real_model
        shape_static = FakeGraph(rank == 0)
        shape_static.det_static = None          # cls / seg graphs accept every batch of their shape: no exchange needed
        runner.graphed['seg'] = shape_static
        assert (runner._graph_for('seg', {}) is not None) == (rank == 0)
        assert runner._graph_for('cls', {}) is None
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_grad_sync_and_packed_scalars_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == 'ok' for r in res), res
