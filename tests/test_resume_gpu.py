"""Checkpoint interop and weight-plane freshness on the GPU.

* save_checkpoint -> resume into a FRESH model and optimizer continues the run bit for bit: the next round's losses, the
  gradient arena after every iteration and the updated weights equal the uninterrupted run's (mmcv layout:
  /root/reference/models/multi/multitask_learner.py:308-353 loads the same layout; mtl/apis/train.py:115-116 resumes).
* After hipGraph replays (which hold the optimizer step) a non-captured forward must see the CURRENT weights through the
  pre-split bf16 weight planes: an evaluation after replays equals the same evaluation after a forced refresh."""
import numpy as np
import pytest
import torch

from util import build_model, load_model_cfg

pytestmark = pytest.mark.gpu


def _runner(cfg, mcfg, seed, cuda, size, graphs):
    from rscotr_amd.data import build_synthetic_multidataloader
    from rscotr_amd.runner import build_runner
    model = build_model(mcfg, seed=seed, perturb=False).to(cuda)
    loader = build_synthetic_multidataloader(cfg, cuda, size=size, batch_size=2, pool=1)
    kw = {} if graphs else dict(graph_tasks=())
    return model, build_runner(model, cfg, loader, logger=lambda m: None, **kw)


def _round(runner):
    """One co-training round from fixed seeds: (log variables, gradient arena) after each of the three iterations."""
    torch.manual_seed(77)
    np.random.seed(77)
    rec = []
    for _ in range(3):
        out = runner.train_iter()
        torch.cuda.synchronize()
        rec.append((runner.last_task, dict(out['log_vars']), runner.optimizer.flat_g.clone()))
    return rec


def test_resumed_run_continues_bit_for_bit(cuda, tmp_path):
    from rscotr_amd.checkpoint import save_checkpoint
    cfg, mcfg = load_model_cfg(tiny=False)
    model_a, run_a = _runner(cfg, mcfg, 1, cuda, 256, graphs=False)
    torch.manual_seed(5)
    np.random.seed(5)
    run_a.run(6)  # two rounds: every task has Adam moments and step counts
    torch.cuda.synchronize()
    path = str(tmp_path / 'iter_6.pth')
    save_checkpoint(path, model_a, run_a.optimizer, meta=dict(iter=run_a.iter))
    want = _round(run_a)
    run_a.optimizer.close()

    model_b, run_b = _runner(cfg, mcfg, 9, cuda, 256, graphs=False)  # different initial weights, empty moments
    assert not torch.equal(model_b.state_dict()['backbone.patch_embed.projection.weight'],
                           model_a.state_dict()['backbone.patch_embed.projection.weight'])
    run_b.resume(path)
    assert run_b.iter == 6
    got = _round(run_b)
    for (ta, la, ga), (tb, lb, gb) in zip(want, got):
        assert ta == tb and list(la) == list(lb)
        for k in la:
            assert la[k] == lb[k], (ta, k, la[k], lb[k])   # bitwise: identical floats
        assert torch.equal(ga, gb), (ta, float((ga - gb).abs().max()))
    assert torch.equal(run_a.optimizer.flat_p, run_b.optimizer.flat_p)
    assert torch.equal(run_a.optimizer.flat_m, run_b.optimizer.flat_m) and torch.equal(run_a.optimizer.flat_v, run_b.optimizer.flat_v)
    assert (run_a.optimizer.steps == run_b.optimizer.steps).all()
    run_b.optimizer.close()


def test_weight_planes_are_fresh_after_graph_replays(cuda, six_term):
    """Graphs on, precision mode 3, planes on — round 4's default and today's RSCOTR_GEMM_H3=0 configuration — at BASELINE
    configs[1] size, where the encoder FFN products (10880 x 256 x 2048) take the pre-split-plane route in training AND in a
    no-grad evaluation."""
    from rscotr_amd import ops, synth
    from rscotr_amd._lib import lib
    assert ops.WPLANES.enabled and lib.rscotr_gemm_get_precision() == 3
    cfg, mcfg = load_model_cfg(tiny=False)
    model, runner = _runner(cfg, mcfg, 3, cuda, 512, graphs=True)
    b = synth.make_batch('seg', 2, 512, seed=123, device=cuda)

    def evaluate():
        model.eval()
        try:
            with torch.no_grad():
                ops.WPLANES.begin('seg')
                p = model.whole_inference_seg(b["img"], b["img_metas"], True)
        finally:
            model.train()
        torch.cuda.synchronize()
        return p.clone()

    with runner.on_stream():
        for _ in range(9):   # eager, capture + first replay, replay
            runner.train_iter()
        assert sorted(runner.graphed) == ['cls', 'det', 'seg']
        e1 = evaluate()
        assert ops.WPLANES.entries, 'the evaluation did not take the weight-plane route: the test does not test'
        for _ in range(3):   # three replays: the parameters move, only the graphs' own plane sets are re-split on the device
            runner.train_iter()
        e2 = evaluate()
        ops.WPLANES.bump()   # force a refresh of every plane set
        e3 = evaluate()
    assert torch.isfinite(e2).all()
    assert not torch.equal(e1, e2)          # the weights did move between the two evaluations
    assert torch.equal(e2, e3), float((e2 - e3).abs().max())
    runner.optimizer.close()
