"""Host-logic check on CPU: the product's module wiring (config -> registry -> MTL ->
train_step, batched Hungarian through the C ABI, packed loss parsing) against the oracle, with
the HIP-backed ops patched to the oracle (no GPU here).  The real parity tests are -m gpu."""
import pytest
import torch

from parity import check_step_pair, run_step_pair
from util import build_model, load_model_cfg, patch_ops_with_oracle


@pytest.fixture(scope='module')
def tiny():
    cfg, mcfg = load_model_cfg(tiny=True)
    return mcfg, build_model(mcfg)


@pytest.mark.parametrize('task', ['cls', 'det', 'seg'])
def test_train_step_matches_oracle(tiny, task, monkeypatch):
    patch_ops_with_oracle(monkeypatch)
    mcfg, model = tiny
    out, oout, rec, orec, P = run_step_pair(model, mcfg, task, 64, seed=3)
    check_step_pair(model, out, oout, rec, orec, P)


@pytest.mark.parametrize('seed', [3, 8])
def test_det_static_path_equals_dynamic_path(tiny, monkeypatch, seed):
    """The shape-static det formulation (padded ground truth, extra masked denoising slots, device-shaped
    assignment) computes what the reference-shaped dynamic path computes: same losses, same gradients,
    same assignment indices — and the two share no target-building code."""
    patch_ops_with_oracle(monkeypatch)
    mcfg, model = tiny
    from rscotr_amd import synth
    batch = synth.make_batch('det', 2, 64, seed=seed)
    rnd = synth.make_rnd(model, batch, seed=seed)
    res = {}
    for mode in (True, False):
        model.bbox_head.static_path = mode
        try:
            model.zero_grad(set_to_none=True)
            rec = {}
            out = model.train_step(dict(batch, rnd=rnd, record=rec))
            out['loss'].backward()
            res[mode] = (out, rec, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        finally:
            model.bbox_head.static_path = True
    (o1, r1, g1), (o2, r2, g2) = res[True], res[False]
    assert list(o1['log_vars']) == list(o2['log_vars'])
    for k, v in o1['log_vars'].items():
        assert abs(v - o2['log_vars'][k]) <= 1e-5 * max(abs(v), 1e-3), k
    assert r1['match'].keys() == r2['match'].keys()
    for k in r1['match']:
        assert (r1['match'][k][0] == r2['match'][k][0]).all() and (r1['match'][k][1] == r2['match'][k][1]).all()
    assert g1.keys() == g2.keys()
    for n in g1:
        assert float((g1[n] - g2[n]).abs().max()) <= 1e-5 * max(float(g2[n].abs().max()), 1e-6) + 1e-7, n
    for a, b in zip(r1['det_outs'], r2['det_outs']):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_det_static_refresh_without_host_ground_truth(tiny, monkeypatch):
    """a captured det iteration refreshes its static tensors key by key (DetStatic.update_into).  A batch
    WITHOUT host copies of its ground truth (any loader other than synth / pipeline) must carry the same tensors as a
    host-packed one — same keys, same values — and must refresh a static built either way."""
    patch_ops_with_oracle(monkeypatch)
    import numpy as np
    from rscotr_amd import synth
    from rscotr_amd.det_head import DetStatic
    mcfg, model = tiny
    head, dev = model.bbox_head, torch.device('cpu')
    b1, b2 = synth.make_batch('det', 2, 64, seed=3), synth.make_batch('det', 2, 64, seed=8)
    mk = lambda b, host, **kw: DetStatic(head, b['gt_bboxes'], b['gt_labels'], b['img_metas'], dev,
                                         gt_host=(b['gt_bboxes_host'], b['gt_labels_host']) if host else None, **kw)
    packed = mk(b1, True)
    caps = dict(gcap=packed.gcap, padcap=packed.padcap)
    plain = mk(b1, False, **caps)
    assert set(plain.t) == set(packed.t) == set(DetStatic.KEYS)
    for k in DetStatic.KEYS:
        a, b = packed.t[k], plain.t[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        assert torch.allclose(a.float(), b.float(), rtol=1e-6, atol=1e-7), k
    # a device-path batch refreshes a static captured from a device-path batch and one captured from a packed batch
    for static in (mk(b1, False, **caps), mk(b1, True, **caps)):
        nxt = mk(b2, False, **caps)
        nxt.update_into(static)
        for k in DetStatic.KEYS:
            assert torch.equal(static.t[k], nxt.t[k]), k
        assert static.counts == nxt.counts
    # and a packed batch refreshes a static captured from a device-path batch
    static, nxt = mk(b1, False, **caps), mk(b2, True, **caps)
    nxt.update_into(static)
    for k in DetStatic.KEYS:
        assert torch.equal(static.t[k], nxt.t[k]), k


def test_log_keys_per_task(tiny, monkeypatch):
    """log_vars naming contract (multitask_learner.py:235-243, dino_head.py:183-232)."""
    patch_ops_with_oracle(monkeypatch)
    mcfg, model = tiny
    out, *_ = run_step_pair(model, mcfg, 'det', 64, seed=5)
    keys = list(out['log_vars'])
    assert keys[:3] == ['det.dior.interm_loss_cls', 'det.dior.interm_loss_bbox', 'det.dior.interm_loss_iou']
    assert len(keys) == 40 and keys[-1] == 'det.dior.loss'
    out, *_ = run_step_pair(model, mcfg, 'seg', 64, seed=5)
    assert list(out['log_vars']) == ['seg.potsdam.seg.loss_ce', 'seg.potsdam.seg.acc_seg', 'seg.potsdam.loss']
    out, *_ = run_step_pair(model, mcfg, 'cls', 64, seed=5)
    assert list(out['log_vars']) == ['cls.resisc.loss']


def test_unused_parameters_per_task(tiny, monkeypatch):
    """SURVEY.md A.7(5,6): cls touches backbone (minus norm0..2) + cls_head only; norm0 never."""
    patch_ops_with_oracle(monkeypatch)
    mcfg, model = tiny
    out, *_ = run_step_pair(model, mcfg, 'cls', 64, seed=1)
    touched = {n.split('.')[0] for n, p in model.named_parameters() if p.grad is not None}
    assert touched == {'backbone', 'cls_head'}
    out, *_ = run_step_pair(model, mcfg, 'seg', 64, seed=1)
    g = {n for n, p in model.named_parameters() if p.grad is not None}
    assert not any(n.startswith('backbone.norm0') for n in g)
    assert any(n.startswith('shared_encoder') for n in g) and any(n.startswith('seg_head') for n in g)
    assert not any(n.startswith(('bbox_head', 'cls_head')) for n in g)


@pytest.mark.parametrize('scheme', [1, 2, 3, 4, 5, 7, 8])
def test_mlvl_cls_head_matches_oracle(monkeypatch, scheme):
    """`MTL_swin-t-...` configs: MlvlClsHead + MlvlClsPixelDecoder (models/multi/cls_head/mlvl_cls_head.py,
    pixel_decoder.py) — the cls step runs neck + shared encoder; every pooling scheme against the oracle.
    Schemes 5-7 are built for 224x224 inputs (4x4 / 7x7 / 14x14 / 28x28 tokens)."""
    from util import MLVL_CFG
    patch_ops_with_oracle(monkeypatch)
    cfg, mcfg = load_model_cfg(tiny=True, path=MLVL_CFG)
    assert mcfg['cls_head']['type'] == 'MlvlClsHead' and mcfg['cls_head']['scheme'] == 2 and mcfg['seg_head']['num_queries'] == 5
    mcfg['cls_head']['scheme'] = scheme
    model = build_model(mcfg, seed=scheme)
    if scheme in (5, 7, 8):  # give the token weighting a non-uniform value so its gradient path is exercised
        with torch.no_grad():
            model.cls_head.out_proj.weight.add_(0.05 * torch.randn(model.cls_head.out_proj.weight.shape))
    size = 224 if scheme in (5, 7) else 64
    out, oout, rec, orec, P = run_step_pair(model, mcfg, 'cls', size, seed=3)
    check_step_pair(model, out, oout, rec, orec, P)
    touched = {n.split('.')[0] for n, p in model.named_parameters() if p.grad is not None}
    assert touched == {'backbone', 'neck', 'shared_encoder', 'cls_head'}
    model.eval()
    from oracle import model as OM
    from rscotr_amd import synth
    b = synth.make_batch('cls', 2, size, seed=2)
    got = model(task='cls', img=b['img'], img_metas=b['img_metas'], return_loss=False)
    import numpy as np
    assert np.allclose(np.stack(got), OM.simple_test(P, mcfg, 'cls', b['img'], b['img_metas']).detach().numpy(), rtol=1e-4, atol=1e-6)


def test_hip_ops_fail_loudly_without_gpu():
    from rscotr_amd import ops
    v = torch.randn(1, 16, 8, 32)
    with pytest.raises(RuntimeError):
        ops.msda(v, torch.tensor([[4, 4]]), torch.tensor([0]), torch.zeros(1, 2, 8, 1, 4, 2), torch.zeros(1, 2, 8, 1, 4))


def test_inference_path_matches_oracle_cpu(tiny, monkeypatch):
    """MTL.forward(return_loss=False) -> simple_test_{cls,det,seg} (multitask_learner.py:91-227) with the HIP-backed ops
    patched to the oracle: host logic of the inference path (top-k decoding, bbox2result, resize / arg-max)."""
    import numpy as np
    from oracle import model as OM
    from rscotr_amd import synth
    from util import state_to_oracle
    patch_ops_with_oracle(monkeypatch)
    mcfg, model = tiny
    model.eval()
    try:
        P = state_to_oracle(model)
        b = synth.make_batch('cls', 2, 64, seed=2)
        out = model(task='cls', img=b['img'], img_metas=b['img_metas'], return_loss=False)
        assert np.allclose(np.stack(out), OM.simple_test(P, mcfg, 'cls', b['img'], b['img_metas']).detach().numpy(), rtol=1e-4, atol=1e-6)
        b = synth.make_batch('det', 2, 64, seed=4)
        out = model(task='det', img=b['img'], img_metas=[dict(m) for m in b['img_metas']], return_loss=False)
        ref = OM.simple_test(P, mcfg, 'det', b['img'], b['img_metas'])
        k = mcfg['test_cfg']['det'].get('max_per_img', mcfg['bbox_head']['num_query'])
        for res, (rb, rl) in zip(out, ref):
            got = np.concatenate(res, 0)
            assert got.shape == (min(k, rb.shape[0]), 5)
            assert np.allclose(np.sort(got[:, 4]), np.sort(rb[:, 4].detach().numpy()), rtol=1e-4, atol=1e-6)
        b = synth.make_batch('seg', 2, 64, seed=6)
        out = model(task='seg', img=b['img'], img_metas=b['img_metas'], return_loss=False)
        ref = OM.simple_test(P, mcfg, 'seg', b['img'], b['img_metas'])
        assert float((np.stack(out) == ref.numpy()).mean()) >= 0.99
    finally:
        model.train()
