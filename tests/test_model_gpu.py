"""Parity of the product path (HIP kernels through the C ABI, on the GPU) against the oracle (CPU)
for one full MTL.train_step per task: losses, log keys, gradients of every parameter, and
bit-exact Hungarian indices.  Tolerance 1e-3 relative (BASELINE.json north_star)."""
import pytest
import torch

from parity import check_step_pair, ranges_checked, run_step_pair
from util import build_model, load_model_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def tiny(cuda):
    cfg, mcfg = load_model_cfg(tiny=True)
    return mcfg, build_model(mcfg).to(cuda)


@pytest.mark.parametrize('task', ['cls', 'det', 'seg'])
def test_train_step_tiny(tiny, task, cuda):
    mcfg, model = tiny
    out, oout, rec, orec, P = run_step_pair(model, mcfg, task, 64, seed=3, device=cuda)
    check_step_pair(model, out, oout, rec, orec, P)


@pytest.mark.parametrize('task', ['cls', 'det', 'seg'])
def test_train_step_main_config_256(task, cuda):
    """The real config (600 queries, 100 CDN) at 256x256, B=2: N = 1360 encoder tokens."""
    cfg, mcfg = load_model_cfg(tiny=False)
    model = build_model(mcfg, seed=1).to(cuda)
    out, oout, rec, orec, P = run_step_pair(model, mcfg, task, 256, seed=11, device=cuda, fp64=True)
    check_step_pair(model, out, oout, rec, orec, P)


def test_cls_step_224_bs1_matches_oracle(cuda):
    """BASELINE configs[0]'s shape (Swin-T, RESISC45 classification, 224 x 224, bs = 1: 3 136 stage-1 tokens, 7 x 7 at
    stage 4) on the HIP path: the cls iteration of the MTL model against the oracle — the reference's CPU-runnable case."""
    cfg, mcfg = load_model_cfg(tiny=False)
    model = build_model(mcfg, seed=5).to(cuda)
    out, oout, rec, orec, P = run_step_pair(model, mcfg, 'cls', 224, seed=13, device=cuda, batch_size=1, fp64=True)
    check_step_pair(model, out, oout, rec, orec, P)


@pytest.mark.parametrize('prec', [0, 3])
@pytest.mark.parametrize('task', ['cls', 'det', 'seg'])
def test_train_step_main_config_512(task, prec, cuda):
    """BASELINE configs[1] itself (512x512, B=2: N = 5440 encoder tokens, 10880-row products) against the oracle under the
    default precision mode of the GEMM: losses, gradients of every parameter (two tiers against the fp32 oracle, element-wise
    bound of the loose tier included; the fp64 anchor of tests/parity.py on top) and the Hungarian indices — under BOTH
    fp32-accurate precision modes of the GEMM: 0 (fp32 matrix pipe) and 3 (bf16x6: three-plane split product)."""
    from rscotr_amd._lib import lib
    old = lib.rscotr_gemm_get_precision()
    lib.call('rscotr_gemm_set_precision', prec)
    try:
        cfg, mcfg = load_model_cfg(tiny=False)
        model = build_model(mcfg, seed=4).to(cuda)
        with ranges_checked() as R:  # (every carried / parameter range word of the iteration verified against its tensor)
            out, oout, rec, orec, P = run_step_pair(model, mcfg, task, 512, seed=17, device=cuda, fp64=True)
            assert prec != 3 or not R.enabled or R.stats.get('checked', 0) > 100
    finally:
        lib.call('rscotr_gemm_set_precision', old)
    check_step_pair(model, out, oout, rec, orec, P)


def test_det_static_path_equals_dynamic_path_full_size(cuda):
    """BASELINE configs[1] size (512x512, B=2, 600 queries, 100 CDN): the shape-static det iteration (padded ground
    truth, masked extra denoising slots, device-side assignment) against the reference-shaped dynamic path (host
    SciPy-exact solver, per-image targets) on the same weights, batch and noise draws: same assignment indices,
    same losses."""
    from rscotr_amd import synth
    cfg, mcfg = load_model_cfg(tiny=False)
    model = build_model(mcfg, seed=2).to(cuda)
    batch = synth.make_batch('det', 2, 512, seed=21, device=cuda)
    rnd = synth.make_rnd(model, synth.make_batch('det', 2, 512, seed=21), seed=21, device=cuda)
    res = {}
    for mode in (True, False):
        model.bbox_head.static_path = mode
        try:
            model.zero_grad(set_to_none=True)
            rec = {}
            out = model.train_step(dict(batch, rnd=rnd, record=rec))
            out['loss'].backward()
            res[mode] = (out, rec)
        finally:
            model.bbox_head.static_path = True
    (o1, r1), (o2, r2) = res[True], res[False]
    assert list(o1['log_vars']) == list(o2['log_vars']) and len(o1['log_vars']) == 40
    assert r1['match'].keys() == r2['match'].keys() and len(r1['match']) == 14
    for k in r1['match']:
        assert (r1['match'][k][0] == r2['match'][k][0]).all() and (r1['match'][k][1] == r2['match'][k][1]).all(), k
    for k, v in o1['log_vars'].items():
        assert abs(v - o2['log_vars'][k]) <= 1e-4 * max(abs(v), 1e-3), (k, v, o2['log_vars'][k])


def test_swin_b_variant_matches_oracle(cuda):
    """BASELINE configs[4] backbone (Swin-B: embed 128, depths 2-2-18-2, heads 4-8-16-32, neck inputs 256/512/1024)
    at a reduced 128x128 input: one seg train step against the oracle (other widths of every kernel: C = 128..1024,
    LayerNorm up to 4096 wide, 32-head windows)."""
    cfg, mcfg = load_model_cfg(tiny=True)
    mcfg['backbone'].update(embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
    mcfg['neck']['in_channels'] = [256, 512, 1024]
    mcfg['cls_head']['in_channels'] = 1024
    model = build_model(mcfg, seed=4).to(cuda)
    out, oout, rec, orec, P = run_step_pair(model, mcfg, 'seg', 128, seed=5, device=cuda)
    check_step_pair(model, out, oout, rec, orec, P)


@pytest.mark.parametrize('scheme,size', [(2, 256), (7, 224), (8, 128)])
def test_mlvl_cls_head_variant(cuda, scheme, size):
    """the `MTL_swin-t-...` configs' MlvlClsHead (cls token from the shared encoder's memories):
    one cls train step on the GPU against the oracle (scheme 2 = the configs' value; 7 / 8 = the learned weightings)."""
    from util import MLVL_CFG
    cfg, mcfg = load_model_cfg(tiny=False, path=MLVL_CFG)
    mcfg['cls_head']['scheme'] = scheme
    model = build_model(mcfg, seed=scheme).to(cuda)
    out, oout, rec, orec, P = run_step_pair(model, mcfg, 'cls', size, seed=13, device=cuda)
    # (scheme 7 at 224^2: every tensor within 1.2 x max(eo, amb), the median inside the step's coin-toss band — the one case of
    # this file where the median gate is the relative one: tests/parity.py, ANCHOR_K_MED)
    check_step_pair(model, out, oout, rec, orec, P, median_rel=scheme == 7)


@pytest.mark.parametrize('task', ['cls', 'det', 'seg'])
def test_train_step_main_config_512_on_the_optimizer_arena(task, cuda):
    """BASELINE configs[1] against the oracle AS THE RUNNER EXECUTES IT: parameters and gradients in FlatAdamW's arenas, the gradient
    sink armed — which is what switches on the routes keyed by a parameter's arena address: weight operands from pre-split fp16
    planes (ops.HPLANES) and the fused two-Linear launches (ops.FFN_FUSED: encoder FFN in det / seg, the detection decoder's FFN, Swin
    stage 1-3 MLPs in all three; the few-row ones as partial sums over runs of the hidden width).  The plain whole-step tests above run
    without an optimizer and never take them."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW, build_param_groups
    cfg, mcfg = load_model_cfg(tiny=False)
    model = build_model(mcfg, seed=4).to(cuda)
    opt = FlatAdamW(build_param_groups(model, dict(type='AdamW', lr=1e-4, weight_decay=0.05)))
    try:
        n0, h0, l0 = ops.FFN_FUSED.calls, len(ops.HPLANES.entries), ops.LIN_FUSED.calls
        with ranges_checked():
            out, oout, rec, orec, P = run_step_pair(model, mcfg, task, 512, seed=17, device=cuda, fp64=True, opt=opt)
        if ops.RANGES.enabled and ops.FFN_FUSED.enabled:
            # Swin stages 1-3: 10 blocks x (forward + backward); the shared encoder: 6 layers x 2 (det, seg); the detection decoder
            # (1600 query rows): 6 layers x 2
            assert ops.FFN_FUSED.calls - n0 == dict(cls=20, det=44, seg=32)[task], ops.FFN_FUSED.calls - n0
            assert len(ops.HPLANES.entries) > h0
            if ops.LIN_FUSED.enabled:  # Swin stages 1-2: 4 blocks x (qkv, proj) x (forward + input gradient), PatchMerging's reduction
                assert ops.LIN_FUSED.calls - l0 >= 16, ops.LIN_FUSED.calls - l0
        check_step_pair(model, out, oout, rec, orec, P)
    finally:
        ops.DEFER.drop()
        opt.close()
