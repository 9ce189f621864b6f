"""Fused clip+AdamW HIP kernels vs torch.optim.AdamW + clip_grad_norm_ (independent primitives),
and a multi-step co-training run (cls, det, seg, cls, ...) against the oracle loop."""
import copy

import pytest
import torch

from util import build_model, load_model_cfg, state_to_oracle

pytestmark = pytest.mark.gpu


def test_fused_adamw_matches_torch(cuda):
    from rscotr_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(33, 7), (1,), (128, 64), (5,), (4097,), (3, 3, 3, 3)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(cuda)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    lrs = [1e-3, 5e-4, 1e-3, 1e-4, 1e-3, 2e-3]
    wds = [1e-2, 0.0, 1e-2, 1e-2, 0.0, 1e-1]
    groups = [dict(name=str(i), param=p, lr=lr, weight_decay=wd) for i, (p, lr, wd) in enumerate(zip(ps, lrs, wds))]
    opt = FlatAdamW(groups, betas=(0.9, 0.999), eps=1e-8, grad_clip=dict(max_norm=0.1, norm_type=2))
    topt = torch.optim.AdamW([dict(params=[p], lr=lr, weight_decay=wd) for p, lr, wd in zip(ref, lrs, wds)],
                             betas=(0.9, 0.999), eps=1e-8, foreach=False)
    for step in range(5):
        opt.zero_grad()
        live = [0, 2, 3, 4, 5] if step == 0 else list(range(6))  # tensor 1 gets its first grad at step 1
        for i in live:
            gr = torch.randn(shapes[i], generator=g).to(cuda) * (10.0 if step % 2 else 0.01)
            ps[i].grad.add_(gr)
            opt.live[i] = True
            ref[i].grad = gr.clone()
        with_grad = [p for p in ref if p.grad is not None]
        norm = torch.nn.utils.clip_grad_norm_(with_grad, 0.1, 2)
        topt.step()
        opt.step()
        assert torch.allclose(opt.grad_norm(), norm, rtol=1e-5)
        for a, b in zip(ps, ref):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), float((a - b).abs().max())


def test_cotraining_steps_match_oracle(cuda):
    """6 iterations of the round-robin loop (zero_grad -> backward -> clip 0.1 -> AdamW) with
    torch-1.11 zero-fill semantics: per-step losses agree with the oracle loop to 1e-3."""
    from oracle import model as OM
    from oracle.optim import OracleOptimizer
    from rscotr_amd import synth
    from rscotr_amd.optim import build_optimizer
    cfg, mcfg = load_model_cfg(tiny=True)
    model = build_model(mcfg).to(cuda)
    P = state_to_oracle(model)
    opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config)
    oopt = OracleOptimizer({k: v for k, v in P.items() if v.requires_grad}, cfg.optimizer, max_norm=0.1)
    for it in range(6):
        task = ('cls', 'det', 'seg')[it % 3]
        b_cpu = synth.make_batch(task, 2, 64, seed=it)
        b_dev = synth.make_batch(task, 2, 64, seed=it, device=cuda)
        rnd = synth.make_rnd(model, b_cpu, seed=it)
        rnd_dev = synth.make_rnd(model, b_cpu, seed=it, device=cuda)
        out = model.train_step(dict(b_dev, rnd=rnd_dev))
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
        oout = OM.train_step(P, mcfg, b_cpu, rnd)
        oopt.zero_grad()
        oout['loss'].backward()
        oopt.step()
        a, b = float(out['loss']), float(oout['loss'])
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-3), (it, task, a, b)
    # parameters that never receive a gradient are never updated (backbone.norm0.*)
    sd = model.state_dict()
    assert torch.equal(sd['backbone.norm0.weight'].cpu(), torch.ones_like(sd['backbone.norm0.weight'].cpu()))
    # weights moved by at most ~lr per step and stay close to the oracle's
    worst = max(float((sd[k].cpu() - P[k].detach()).abs().max()) for k in P if P[k].requires_grad)
    assert worst <= 6 * 5e-5 * 1.5, worst


def test_graphed_iterations_match_eager(cuda):
    """hipGraph replay of the cls / seg iterations (rscotr_amd.runner.GraphedTask) vs the eager loop
    on the same batch sequence: same losses and the same weights afterwards."""
    import copy
    from rscotr_amd import synth
    from rscotr_amd.runner import IterBasedRunner
    from rscotr_amd.optim import build_optimizer
    cfg, mcfg = load_model_cfg(tiny=True)
    mcfg = copy.deepcopy(mcfg)
    mcfg['backbone']['drop_path_rate'] = 0.0  # DropPath draws differ between captured and eager RNG streams
    tasks = ['cls', 'seg', 'cls', 'seg', 'cls', 'seg']
    batches = [synth.make_batch(t, 2, 64, seed=40 + i, device=cuda) for i, t in enumerate(tasks)]

    def run(graph_tasks, seq):
        model = build_model(mcfg).to(cuda)
        model.cls_augments.draw = lambda *a, **k: dict(kind='identity')
        opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config)
        r = IterBasedRunner(model, opt, [dict(b, img_metas=[dict(m) for m in b['img_metas']]) for b in seq],
                            graph_tasks=graph_tasks)
        logs = [r.train_iter()['log_vars'] for _ in seq]
        opt.close()
        return model, logs

    # the capturing iteration applies its batch ONCE (the two warm-up runs are rolled back): the graphed run walks the same
    # trajectory as the eager loop, batch for batch
    m_e, logs_e = run((), batches)
    m_g, logs_g = run(('cls', 'seg'), batches)
    for a, b in ((logs_g[2], logs_e[2]), (logs_g[3], logs_e[3]), (logs_g[4], logs_e[4]), (logs_g[5], logs_e[5])):
        assert list(a) == list(b)
        for k in a:
            if 'loss' not in k:
                continue  # acc_seg is an arg-max statistic of a near-random tiny model
            # (seg: the loss passes through 9 layers of hard `sigmoid(mask) < 0.5` attention masks)
            tol = 2e-3 if k.startswith('cls') else 1e-2
            assert abs(a[k] - b[k]) <= tol * max(abs(b[k]), 1e-3), (k, a[k], b[k])
    sd_e, sd_g = m_e.state_dict(), m_g.state_dict()
    # AdamW moves every weight by ~lr per step whatever the gradient's size, so a near-zero gradient whose
    # sign is noise can diverge by 2*lr per step; on average the two runs must coincide
    diffs = torch.cat([(sd_e[k] - sd_g[k]).abs().flatten() for k in sd_e if sd_e[k].dtype.is_floating_point])
    assert float(diffs.max()) <= 6 * 2 * 5e-5, float(diffs.max())
    assert float(diffs.mean()) <= 2e-5, float(diffs.mean())  # a dropped update would show as ~6 * lr


@pytest.mark.parametrize('task', ['seg', 'det'])
def test_deferred_splitk_combine_matches_immediate(cuda, task):
    """The split-K weight gradients of one backward pass combined by ONE launch at the end (ops.DEFER) against the
    combine launched with each contraction: same gradient arena (fixed summation order in both: agreement to fp32
    rounding of the partial sums), at a size whose token count makes the reductions split (256x256: K = 2720)."""
    from rscotr_amd import ops, synth
    from rscotr_amd.optim import build_optimizer
    cfg, mcfg = load_model_cfg(tiny=False)
    model = build_model(mcfg, seed=3).to(cuda)
    opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config)
    batch = synth.make_batch(task, 2, 256, seed=5, device=cuda)
    rnd = synth.make_rnd(model, synth.make_batch(task, 2, 256, seed=5), seed=5, device=cuda)
    res = {}
    try:
        for mode in (True, False):
            ops.DEFER.enabled = mode
            opt.zero_grad()
            out = model.train_step(dict(batch, rnd=rnd))
            out['loss'].backward()
            pending = len(ops.DEFER.entries) + len(ops.DEFER.group)  # split-K slabs waiting for the combine + grouped problems
            assert (pending > 20) == mode, pending
            ops.flush_deferred()
            assert not ops.DEFER.pending() and not ops.DEFER.notify
            res[mode] = (opt.flat_g.clone(), float(opt.grad_norm() if False else opt.flat_g.norm()))
    finally:
        ops.DEFER.enabled = True
        opt.close()
    a, b = res[True][0], res[False][0]
    assert float(b.abs().max()) > 0
    # The two passes differ by the (fixed) summation order of up to 64 slabs and by the run-to-run order of the few fp32
    # atomics of the backward pass (MSDA chunk combine, GroupNorm / loss sums): ~1e-5 relative.  A lost or doubled slab
    # would move one tensor by >= 1 / 64: every parameter's gradient is compared on its own.
    assert float((a - b).norm() / b.norm()) <= 1e-4
    gn = float(b.norm())
    for g_, o in zip(opt.groups, opt.offsets):
        n = g_['param'].numel()
        nb, nd = float(b[o:o + n].norm()), float((a[o:o + n] - b[o:o + n]).norm())
        assert nd <= 2e-3 * nb + 1e-5 * gn, (g_['name'], nd, nb)  # (tensors with a near-zero gradient only hold the noise)
