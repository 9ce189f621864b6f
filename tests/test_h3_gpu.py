"""The fp16 split product ("h3", rscotr_gemm_f32_r) and the value-range words it scales its operands with: rscotr_amax_f32 /
rscotr_amax_group / the optimizer's parameter words / the producers' amax_out — against fp64 and against torch's max |x|."""
import contextlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.double()
    return float((a.detach().cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))


class _Binade:
    """What a range word says about a tensor's maximum: its binade [lo, hi) (csrc/common.h: the word is an exponent map).  Compares equal
    to the maxima that lie in it, so `_word(ops, slot) == float(t.abs().max())` reads as before."""

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def __eq__(self, m):
        return (self.lo <= m < self.hi) or (m == 0.0 and self.hi == 0.0)

    def __repr__(self):
        return f'[{self.lo}, {self.hi})'


def _word(ops, slot):
    """binade of a range word"""
    return _Binade(*ops.RANGES.word(slot))


@contextlib.contextmanager
def ranges_on(ops, check=False):
    """the opt-in route for the duration of a test (restored on exit: the suite tests the shipped default elsewhere)"""
    old = (ops.RANGES.enabled, ops.RANGES.check, ops.WPLANES.enabled)
    ops.RANGES.enabled, ops.RANGES.check, ops.WPLANES.enabled = True, check, False
    try:
        yield ops.RANGES
    finally:
        ops.RANGES.enabled, ops.RANGES.check, ops.WPLANES.enabled = old


@pytest.mark.parametrize('shape,ld', [((1, 1), 1), ((37, 45), 45), ((200, 256), 256), ((1000, 96), 128), ((32768, 288), 288),
                                      ((10880, 2048), 2048), ((5, 7), 9)])
def test_amax_word_equals_max_abs(cuda, shape, ld):
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    g = torch.Generator().manual_seed(shape[0] + shape[1])
    full = (torch.randn((shape[0], ld), generator=g) * 3.0).to(cuda)
    full[:, shape[1]:] = 1e9  # (columns past `cols` of a strided operand must not count)
    slot = ops.RANGES.new_slot(cuda)
    lib.call('rscotr_amax_f32', full.data_ptr(), shape[0], shape[1], ld, slot, torch.cuda.current_stream().cuda_stream)
    assert _word(ops, slot) == float(full[:, :shape[1]].abs().max())


def test_amax_group_measures_many_tensors_in_one_launch(cuda):
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    shapes = [(512, 768), (2048, 1152), (8192, 576), (32768, 96), (32768, 48), (200, 256), (37, 45), (1, 4)]
    ts = [torch.randn(s, device=cuda) * (i + 1) for i, s in enumerate(shapes)]
    rows, first, slots = [], 0, []
    for t in ts:
        slots.append(ops.RANGES.new_slot(cuda))
        rows.append((t.data_ptr(), t.shape[0], t.shape[1], t.shape[1], slots[-1], first))
        first += max(1, min(128, t.numel() // 65536))
    tab = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(cuda)
    lib.call('rscotr_amax_group', tab.data_ptr(), len(rows), first, torch.cuda.current_stream().cuda_stream)
    for t, s in zip(ts, slots):
        assert _word(ops, s) == float(t.abs().max())


@pytest.mark.parametrize('M,N,K,ak,bk', [(10880, 2048, 256, 0, 0), (10880, 256, 2048, 0, 1), (10880, 256, 256, 0, 0),
                                         (2048, 1536, 384, 0, 0), (8192, 192, 768, 0, 1), (256, 2048, 10880, 1, 1),
                                         (384, 1536, 2048, 1, 1), (4096, 4096, 4096, 0, 0), (1000, 768, 3072, 0, 0),
                                         (53176, 256, 256, 0, 1), (256, 2048, 53176, 1, 1), (32768, 96, 384, 0, 0),
                                         (2000, 1024, 512, 1, 0), (8192, 200, 512, 0, 1), (1604, 2048, 256, 0, 0)])
@pytest.mark.parametrize('sa,sb,tail', [(1.0, 0.05, False), (3e-7, 40.0, False), (1.0, 0.05, True)])
def test_gemm_h3_is_fp32_accurate(cuda, gemm_precision, six_term, M, N, K, ak, bk, sa, sb, tail):
    """Two fp16 planes per operand, power-of-two scales from the range words, three MFMAs per k-step: against fp64 next to
    the fp32 matrix pipe on the shapes precision mode 3 routes to the split kernels (all four layouts, both tile sizes, edge
    instantiations, k-slices, fused epilogue).  Same bound as the six-term bf16 product (tests/test_gemm_gpu.py): at most
    1e-6 of max |C| or 1.5 x the fp32 pipe's own error up to K = 2048, never more than twice the fp32 pipe's + 5e-7 — also for
    operands far from 1 in magnitude (3e-7: gradients; 40: un-normalised activations) and for a heavy-tailed operand (one
    entry in ~10 000 is 10 000 x the rest: the small entries must keep their relative precision next to the outliers)."""
    from rscotr_amd import ops
    if (tail or sa != 1.0) and float(M) * N * K > 8e9:
        pytest.skip('scaled / heavy-tail operands on the smaller shapes only (fp64 reference time)')
    g = torch.Generator().manual_seed(M + N + K + ak + bk)
    A = torch.randn((K, M) if ak else (M, K), generator=g) * sa
    B = torch.randn((K, N) if bk else (N, K), generator=g) * sb
    if tail:
        A.view(-1)[::9973] *= 1e4
    bias, resid = torch.randn(N, generator=g) * sa * sb, torch.randn(M, N, generator=g) * sa * sb
    ref = ((A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t()) + bias.double()).clamp(min=0) \
        + resid.double()
    Ad, Bd, biasd, residd = A.to(cuda), B.to(cuda), bias.to(cuda), resid.to(cuda)
    gemm_precision(0)
    o32 = ops.gemm(Ad, Bd, M, N, K, A.shape[1], B.shape[1], ak, bk, bias=biasd, act=1, resid=residd)
    gemm_precision(3)
    o6 = ops.gemm(Ad, Bd, M, N, K, A.shape[1], B.shape[1], ak, bk, bias=biasd, act=1, resid=residd)
    with ranges_on(ops) as R:
        before = dict(R.stats)
        oh = ops.gemm(Ad, Bd, M, N, K, A.shape[1], B.shape[1], ak, bk, bias=biasd, act=1, resid=residd)
        assert R.stats['measured'] == before['measured'] + 2, 'the product did not ask for its operands\' ranges'
        # the epilogue left the range of what it stored
        assert _word(ops, R.slot_of(oh)) == float(oh.abs().max())
    e32, e6, eh = _rel(o32, ref), _rel(o6, ref), _rel(oh, ref)
    assert not torch.equal(oh, o6), 'the fp16 split product was not taken'
    assert eh <= 2.0 * e32 + 5e-7, (e32, e6, eh)
    if K <= 2048:
        assert eh <= max(1e-6, 1.5 * e32), (e32, e6, eh)
    if M % 64 or N % 64:
        guard = torch.full((M + 8, N), 7.0, device=cuda)
        with ranges_on(ops):
            ops.gemm(Ad, Bd, M, N, K, A.shape[1], B.shape[1], ak, bk, out=guard[:M], bias=biasd, act=1, resid=residd)
        assert torch.equal(guard[:M], oh) and bool((guard[M:] == 7.0).all())


def test_gemm_h3_zero_and_tiny_operands(cuda):
    """An all-zero operand (range word 0: the scale clamps) and magnitudes near the bottom of fp32's range stay finite and
    exact to the usual bound."""
    from rscotr_amd import ops
    M, N, K = 2048, 384, 512
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(cuda)
    B = torch.randn(N, K, generator=g).to(cuda)
    with ranges_on(ops):
        z = ops.gemm(torch.zeros_like(A), B, M, N, K, K, K, 0, 0)
        assert bool((z == 0).all())
        tiny = ops.gemm(A * 1e-30, B * 1e-5, M, N, K, K, K, 0, 0)
    ref = (A.double() * 1e-30) @ (B.double() * 1e-5).t()
    assert torch.isfinite(tiny).all() and _rel(tiny, ref.cpu()) < 2e-6


def test_h3_weight_gradient_routes(cuda):
    """dW with the bias gradient riding along, per-sample k scaling and accumulation into C — immediate combine, deferred
    slabs and the grouped launch's split-product variant (whose operands' ranges are measured by ONE grouped launch)."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(11)
    Mt, N, K = 10880, 256, 2048   # tokens, dy columns, x columns
    dy = (torch.randn(Mt, N, generator=g) * 1e-4).to(cuda)
    x = torch.randn(Mt, K, generator=g).to(cuda)
    ks = (torch.rand(2, generator=g) + 0.5).to(cuda)
    per = Mt // 2
    scaled = dy.double().cpu() * ks.double().cpu().repeat_interleave(per)[:, None]
    ref = scaled.t() @ x.double().cpu()
    with ranges_on(ops, check=True):
        rs = torch.zeros(N, device=cuda)
        out = ops.gemm(dy, x, N, K, Mt, N, K, 1, 1, rowsum=rs, kscale=ks, krows_per=per)
    assert _rel(out, ref) < 2e-6 and _rel(rs, scaled.sum(0)) < 2e-6


def test_grouped_weight_gradients_on_the_fp16_body(cuda):
    """rscotr_gemm_dw_group variant 7 through ops.DEFER: small-output weight gradients of one backward pass in ONE launch on the
    fp16 split product, the operands' range words measured by ONE grouped launch (rscotr_amax_group) right before it."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(256, 256, 10880), (256, 256, 1600), (96, 48, 4096), (384, 384, 2048), (192, 576, 8192), (128, 256, 10880)]
    outs = [torch.nn.Parameter(torch.zeros(M, N, device=cuda)) for M, N, K in shapes]   # destinations in a gradient arena
    opt = FlatAdamW([dict(name=f'w{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(outs)])
    try:
        with ranges_on(ops) as R:
            R.stats = {k: 0 for k in R.stats}
            want, keep = [], []
            for (M, N, K), p in zip(shapes, outs):
                A = (torch.randn(K, M, generator=g) * 1e-3).to(cuda)
                B = torch.randn(K, N, generator=g).to(cuda)
                keep.append((A, B))
                want.append(A.double().cpu().t() @ B.double().cpu())
                ops.gemm(A, B, M, N, K, M, N, 1, 1, out=p.grad, accumulate=True)
            assert ops.DEFER.group and all(e[11] and e[12] for e in ops.DEFER.group if min(e[5], e[6]) >= 48)
            ops.flush_deferred()
            assert R.stats.get('grouped', 0) == 2 * len(shapes)
        torch.cuda.synchronize()
        for p, w in zip(outs, want):
            assert _rel(p.grad, w) < 2e-6, tuple(p.shape)
    finally:
        opt.close()


def test_optimizer_keeps_parameter_ranges(cuda):
    """FlatAdamW.amax_slot: the word of every parameter equals max |w| at construction and after update steps (refreshed by the
    update kernel itself for the tensors it steps, untouched for the others)."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, device=cuda) * (i + 1)) for i, s in enumerate([(300, 40), (7,), (4096, 33), (5000,)])]
    groups = [dict(name=f'p{i}', param=p, lr=1e-2, weight_decay=0.05) for i, p in enumerate(ps)]
    opt = FlatAdamW(groups, grad_clip=dict(max_norm=0.1))
    try:
        for p in ps:
            assert _word(ops, opt.amax_slot(p.data_ptr())) == float(p.detach().abs().max())
        for step in range(3):
            opt.zero_grad()
            for i, p in enumerate(ps[:3]):  # the last tensor never receives a gradient: never stepped, word untouched
                p.grad.copy_(torch.randn_like(p))
                opt._on_ready(i)
            opt.step()
            torch.cuda.synchronize()
            for p in ps:
                assert _word(ops, opt.amax_slot(p.data_ptr())) == float(p.detach().abs().max()), step
        # a slice of a parameter is bounded by the parameter's word
        assert opt.amax_slot(ps[2].data_ptr() + 4 * 1000) == opt.amax_slot(ps[2].data_ptr())
    finally:
        opt.close()


@pytest.mark.parametrize('seed', [4, 23])
@pytest.mark.parametrize('task', ['cls', 'det', 'seg'])
def test_train_step_with_the_fp16_split_product(cuda, task, seed):
    """The whole co-training step with the opt-in route on, every carried range word verified against the tensor it travels
    with (RANGES.check) — parity with the oracle as in tests/test_model_gpu.py."""
    from parity import check_step_pair, run_step_pair
    from rscotr_amd import ops
    from util import build_model, load_model_cfg
    cfg, mcfg = load_model_cfg()
    model = build_model(mcfg).to(cuda)
    with ranges_on(ops, check=True) as R:
        R.stats = {k: 0 for k in R.stats}
        out, oout, rec, orec, P = run_step_pair(model, mcfg, task, 256, seed=seed, device=cuda)
        assert R.stats['carried'] > 0 and R.stats.get('checked', 0) > 0
    check_step_pair(model, out, oout, rec, orec, P)


def _mlp_grads(ops, x, w1, b1, w2, b2, dy, begin_between):
    x = x.clone().requires_grad_(True)
    ops.RANGES.begin(x.device)
    y = ops.mlp(x, [(w1, b1), (w2, b2)], act='relu', identity=x)
    if begin_between:
        ops.RANGES.begin(x.device)  # a second forward / an evaluation between forward and backward starts a new generation
        junk = torch.full((64, 64), 1e-30, device=x.device)
        for _ in range(8):  # ... and hands the first slots to other tensors
            ops.RANGES.of(junk, 64, 64, 64)
            junk = junk.clone()
    y.backward(dy)
    ops.flush_deferred()
    return [x.grad] + [p.grad.clone() for p in (w1, b1, w2, b2)]


def test_range_words_do_not_outlive_their_generation(cuda):
    """slots carried through an autograd ctx were re-stamped with the generation current at backward time.
    With a begin() between forward and backward the words are zero or someone else's: a zero word scales the operand by 2^116
    and the gradients came back inf / NaN.  They must equal the undisturbed run's (a measured range may differ from a carried
    upper bound by the power of two it rounds to: the products agree to fp32 rounding, not bit for bit)."""
    from rscotr_amd import ops
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    g = torch.Generator().manual_seed(5)
    M, C, H = 2176, 256, 1024
    x = torch.randn(M, C, generator=g).to(cuda)
    dy = torch.randn(M, C, generator=g).to(cuda)
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.05).to(cuda)) for s in ((H, C), (H,), (C, H), (C,))]
    res = {}
    for between in (False, True):
        for p in ps:
            p.grad = None
        res[between] = _mlp_grads(ops, x, *ps, dy, between)
    for a, b in zip(res[True], res[False]):
        assert torch.isfinite(a).all()
        assert _rel(a, b.cpu()) < 2e-6


def test_pending_grouped_gradients_survive_a_new_generation(cuda):
    """Gradient accumulation: the grouped weight-gradient problems of a first backward are still pending (their operands' slot
    addresses held raw by DEFER) when the second forward's begin() zeroes the words."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    g = torch.Generator().manual_seed(6)
    M, C, H = 2176, 256, 512
    xs = [torch.randn(M, C, generator=g).to(cuda) for _ in range(2)]
    dys = [torch.randn(M, C, generator=g).to(cuda) for _ in range(2)]
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.05).to(cuda)) for s in ((H, C), (H,), (C, H), (C,))]
    ref = [torch.zeros_like(p) for p in ps]
    for x, dy in zip(xs, dys):  # the reference: fp64 on the host
        xd = x.double().cpu()
        h = torch.relu(xd @ ps[0].detach().double().cpu().T + ps[1].detach().double().cpu())
        gy = dy.double().cpu()
        gh = (gy @ ps[2].detach().double().cpu()) * (h > 0)
        for r, v in zip(ref, (gh.T @ xd, gh.sum(0), gy.T @ h, gy.sum(0))):
            r += v.float().to(cuda)
    opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
    try:
        opt.zero_grad()
        for x, dy in zip(xs, dys):
            ops.RANGES.begin(cuda)  # (MTL.forward does this at the start of every iteration)
            y = ops.mlp(x, [(ps[0], ps[1]), (ps[2], ps[3])], act='relu')
            y.backward(dy)  # no flush in between: the second begin() finds the first pass's problems pending
        ops.flush_deferred()
        torch.cuda.synchronize()
        for p, r in zip(ps, ref):
            assert torch.isfinite(p.grad).all()
            assert _rel(p.grad, r.cpu()) < 2e-6
    finally:
        ops.DEFER.drop()
        opt.close()


def test_parameter_words_follow_every_writer_of_the_arena(cuda):
    """only MTL.load_state_dict and FlatAdamW.restore marked the parameters' range words stale.  A submodule
    load (nn.Module recursion never calls a child's load_state_dict), init_weights() and load_checkpoint leave them stale too —
    and a stale-small word overflows the fp16 planes."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW, build_param_groups
    from util import build_model, load_model_cfg
    cfg, mcfg = load_model_cfg(tiny=True)
    model = build_model(mcfg).to(cuda)
    opt = FlatAdamW(build_param_groups(model, dict(type='AdamW', lr=1e-3, weight_decay=0.0)))
    try:
        w = model.cls_head.fc.weight if hasattr(model.cls_head, 'fc') else next(model.cls_head.parameters())
        assert _word(ops, opt.amax_slot(w.data_ptr())) == float(w.detach().abs().max())
        sd = {k: v * 64.0 for k, v in model.cls_head.state_dict().items()}
        model.cls_head.load_state_dict(sd)  # a SUBMODULE load
        assert _word(ops, opt.amax_slot(w.data_ptr())) == float(w.detach().abs().max())
        model.init_weights()
        for p in model.parameters():
            if p.requires_grad and p.numel():
                assert _word(ops, opt.amax_slot(p.data_ptr())) == float(p.detach().abs().max())
    finally:
        opt.close()
