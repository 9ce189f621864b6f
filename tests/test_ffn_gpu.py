"""The encoder FFN as one launch per direction (rscotr_ffn_h3, csrc/ffn.hip; ops.FFN_FUSED) against the two-product route it
replaces and against fp64: hidden tensor, outputs and gradients at fp32 rounding, ragged row counts, loose range bounds."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.double()
    return float((a.detach().cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))


def _setup(cuda, M, C, H, seed, bias_scale=0.5):
    from rscotr_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C, generator=g).to(cuda)
    dy = torch.randn(M, C, generator=g).to(cuda)
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * sc).to(cuda))
          for s, sc in (((H, C), 0.06), ((H,), bias_scale), ((C, H), 0.03), ((C,), bias_scale))]
    opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
    return x, dy, ps, opt


def _ref64(x, dy, ps, identity):
    xd = x.double().cpu()
    w1, b1, w2, b2 = [p.detach().double().cpu() for p in ps]
    pre = xd @ w1.T + b1
    h = torch.relu(pre)
    y = h @ w2.T + b2 + (xd if identity else 0)
    gy = dy.double().cpu()
    gh = (gy @ w2) * (pre > 0)
    dx = gh @ w1 + (gy if identity else 0)
    return y, dx, (gh.T @ xd, gh.sum(0), gy.T @ h, gy.sum(0)), h


def _run(ops, x, dy, ps, identity, fused):
    old = ops.FFN_FUSED.enabled
    ops.FFN_FUSED.enabled = fused
    try:
        for p in ps:
            p.grad.zero_()
        xx = x.clone().requires_grad_(True)
        ops.RANGES.begin(x.device)
        n0 = ops.FFN_FUSED.calls
        y = ops.mlp(xx, [(ps[0], ps[1]), (ps[2], ps[3])], act='relu', identity=xx if identity else None)
        y.backward(dy)
        ops.flush_deferred()
        torch.cuda.synchronize()
        assert (ops.FFN_FUSED.calls - n0 == 2) == fused, 'route not taken' if fused else 'fused route taken although off'
        return y.detach(), xx.grad.detach(), [p.grad.detach().clone() for p in ps]
    finally:
        ops.FFN_FUSED.enabled = old


@pytest.mark.parametrize('M,H,identity', [(10880, 2048, True), (4352, 2048, False), (2200, 1024, True), (2049, 256, True), (12300, 128, False),
                                          (1600, 2048, True), (1100, 2048, False), (1031, 768, True)])
def test_fused_ffn_matches_fp64_and_the_two_product_route(cuda, M, H, identity):
    """(1600 / 1100 x 2048 and 2200 x 1024 are FEW-ROW launches: the hidden width cut into 4 / 8 / 4 runs whose partial outputs a second
    launch adds up, rscotr_ffn_h3_splits; 1031 x 768 has too few chunks to cut.)"""
    from rscotr_amd import ops
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    C = 256
    x, dy, ps, opt = _setup(cuda, M, C, H, seed=M + H)
    try:
        y64, dx64, gp64, _ = _ref64(x, dy, ps, identity)
        yf, dxf, gpf = _run(ops, x, dy, ps, identity, True)
        yu, dxu, gpu = _run(ops, x, dy, ps, identity, False)
        errs = []
        for name, got, un, ref in [('y', yf, yu, y64), ('dx', dxf, dxu, dx64)] + [(f'dp{i}', a, b, r) for i, (a, b, r) in enumerate(zip(gpf, gpu, gp64))]:
            assert torch.isfinite(got).all()
            errs.append((name, _rel(got, ref), _rel(un, ref)))
        # fp32-FMA-class error: the fused route within the bound the routed products are held to (tests/test_h3_gpu.py)
        assert all(e_f <= max(1e-6, 1.5 * e_u) for _, e_f, e_u in errs), errs
    finally:
        ops.DEFER.drop()
        opt.close()


@pytest.mark.parametrize('M', [10880, 1600])
def test_fused_ffn_hidden_and_range_words(cuda, M):
    """What the fused launch leaves for the weight gradients: the hidden tensor (and the gated dH of the mirrored call) at
    fp32-product accuracy against fp64, the SAME gate as the two-product route takes wherever the pre-activation is not within
    rounding of zero, and range words that hold the binades of the true maxima."""
    from rscotr_amd import ops
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    C, H = 256, 2048
    assert int(ops.lib.rscotr_ffn_h3_splits(M, C, H)) == (1 if M == 10880 else 4)
    x, dy, ps, opt = _setup(cuda, M, C, H, seed=3)
    try:
        ops.RANGES.begin(cuda)
        W1, b1, W2, b2 = [p.data for p in ps]
        bits = torch.empty(int(ops.lib.rscotr_ffn_h3_bits_words(M, C, H)), dtype=torch.int32, device=cuda)
        hid, y = ops.FFN_FUSED.run(x, W1, b1, W2, b2, ops.ACT_RELU, bits, 0, None, True)
        pre64 = x.double() @ W1.double().T + b1.double()
        h64 = torch.relu(pre64)
        assert float((hid.double() - h64).abs().max() / h64.abs().max()) < 1e-6
        sure = pre64.abs() > 1e-5 * float(h64.abs().max())
        assert bool(((hid > 0) == (pre64 > 0))[sure].all())
        for t in (hid, y):  # (a word is the binade of the maximum: csrc/common.h)
            lo, hi = ops.RANGES.word(ops.RANGES.slot_of(t))
            assert lo <= float(t.abs().max()) < hi
        dH, dx = ops.FFN_FUSED.run(dy, W2, None, W1, None, ops.ACT_RELU, bits, 1, None, False)
        dH64 = (dy.double() @ W2.double()) * (hid > 0)  # gated by the bits the forward left = [hid > 0]
        assert float((dH.double() - dH64).abs().max() / dH64.abs().max()) < 1e-6
        assert bool(((dH != 0) <= (hid > 0)).all())
    finally:
        opt.close()


def test_fused_ffn_with_a_loose_bound_keeps_fp32_accuracy(cuda):
    """The hidden planes are scaled from C max|x| max|W1| + max|b1|, not from the hidden tensor's own maximum: one outlier in x
    and in W1 makes the bound ~2^13 too loose for everything else — the second product must stay at fp32-class error."""
    from rscotr_amd import ops
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    M, C, H = 4352, 256, 512
    x, dy, ps, opt = _setup(cuda, M, C, H, seed=9, bias_scale=0.01)
    try:
        x[5, 7] = 90.0
        ps[0].data[3, 100] = 6.0
        opt.params_changed()
        y64, dx64, gp64, _ = _ref64(x, dy, ps, True)
        yf, dxf, gpf = _run(ops, x, dy, ps, True, True)
        for got, ref in [(yf, y64), (dxf, dx64)] + list(zip(gpf, gp64)):
            assert _rel(got, ref) <= 1e-6
    finally:
        ops.DEFER.drop()
        opt.close()


def _gelu64(t):
    return 0.5 * t * (1.0 + torch.erf(t * 0.7071067811865476))


def _gelu_grad64(t):
    return 0.5 * (1.0 + torch.erf(t * 0.7071067811865476)) + t * torch.exp(-0.5 * t * t) * 0.3989422804014327


@pytest.mark.parametrize('B,L,C,drop', [(2, 16384, 96, True), (2, 4096, 192, True), (2, 4096, 192, False), (1, 2500, 96, True), (3, 1000, 192, True),
                                        (1, 4100, 128, True), (2, 1024, 384, True), (2, 1024, 384, False), (3, 500, 384, True), (2, 1024, 192, True)])
def test_fused_swin_mlp_matches_fp64_and_the_two_product_route(cuda, B, L, C, drop):
    """The MLP of a Swin block (Linear - GELU - Linear, H = 4 C, DropPath folded in as a per-sample factor, the identity a separate
    tensor) on the fused route: output, input gradient and the four parameter gradients against fp64 and against the
    two-product route, ragged row counts included.  C = 384 (stage 3 of Swin-T: 2048 rows) and the 2048-row case at C = 192 run as
    partial sums over runs of the hidden width."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    H = 4 * C
    g = torch.Generator().manual_seed(B * L + C)
    x = torch.randn(B, L, C, generator=g).to(cuda)
    ident = torch.randn(B, L, C, generator=g).to(cuda)
    dy = torch.randn(B, L, C, generator=g).to(cuda)
    scale = (torch.tensor([1.25, 0.0, 1.25][:B]) if drop else None)
    ps = [torch.nn.Parameter((torch.randn(sh, generator=g) * sc).to(cuda)) for sh, sc in (((H, C), 0.1), ((H,), 0.3), ((C, H), 0.05), ((C,), 0.3))]
    opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
    old = ops.FFN_FUSED.enabled
    try:
        xd, w1, b1, w2, b2 = x.double().cpu(), *[p.detach().double().cpu() for p in ps]
        sd = torch.ones(B, dtype=torch.float64) if scale is None else scale.double()
        pre = xd @ w1.T + b1
        h = _gelu64(pre)
        y64 = (h @ w2.T + b2) * sd[:, None, None] + ident.double().cpu()
        gy = dy.double().cpu() * sd[:, None, None]
        gh = (gy @ w2) * _gelu_grad64(pre)
        ref = [y64, gh @ w1, gh.reshape(-1, H).T @ xd.reshape(-1, C), gh.sum((0, 1)), gy.reshape(-1, C).T @ h.reshape(-1, H), gy.sum((0, 1))]
        res = {}
        for fused in (True, False):
            ops.FFN_FUSED.enabled = fused
            for p in ps:
                p.grad.zero_()
            xx = x.clone().requires_grad_(True)
            ops.RANGES.begin(cuda)
            n0 = ops.FFN_FUSED.calls
            y = ops.mlp(xx, [(ps[0], ps[1]), (ps[2], ps[3])], act='gelu', identity=ident, out_scale=None if scale is None else scale.to(cuda))
            y.backward(dy)
            ops.flush_deferred()
            torch.cuda.synchronize()
            assert (ops.FFN_FUSED.calls - n0 == 2) == fused
            res[fused] = [y.detach(), xx.grad.detach()] + [p.grad.detach().clone() for p in ps]
        errs = [(i, _rel(a, r), _rel(b, r)) for i, (a, b, r) in enumerate(zip(res[True], res[False], ref))]
        assert all(torch.isfinite(a).all() for a in res[True])
        assert all(e_f <= max(1e-6, 1.5 * e_u) for _, e_f, e_u in errs), errs
    finally:
        ops.FFN_FUSED.enabled = old
        ops.DEFER.drop()
        opt.close()


@pytest.mark.parametrize('B,L,C,drop', [(2, 16384, 96, True), (2, 4096, 192, False), (2, 1024, 384, True), (3, 500, 384, False), (1, 2500, 96, False)])
def test_fused_swin_mlp_with_the_norm_in_front(cuda, B, L, C, drop):
    """x + DropPath(MLP(LayerNorm(x))) with the norm left to the MLP call (ops.layer_norm_fork(lazy=True) -> rscotr_ffn_h3_ln: the rows
    are normalised while the fused kernel stages them): output, the norm's output and statistics, the input gradient and all six
    parameter gradients against fp64 and against the same block with the norm as its own launch."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    H = 4 * C
    g = torch.Generator().manual_seed(B * L + C + 1)
    x = (torch.randn(B, L, C, generator=g) * 1.7 + 0.3).to(cuda)
    dy = torch.randn(B, L, C, generator=g).to(cuda)
    scale = (torch.tensor([1.25, 0.0, 1.25][:B]) if drop else None)
    shapes = (((C,), 0.4, 1.0), ((C,), 0.3, 0.0), ((H, C), 0.1, 0.0), ((H,), 0.3, 0.0), ((C, H), 0.05, 0.0), ((C,), 0.3, 0.0))
    ps = [torch.nn.Parameter((torch.randn(sh, generator=g) * sc + off).to(cuda)) for sh, sc, off in shapes]
    opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
    old = ops.FFN_FUSED.ln
    try:
        xd = x.double().cpu()
        gm, bt, w1, b1, w2, b2 = [p.detach().double().cpu().requires_grad_(True) for p in ps]
        xr = xd.clone().requires_grad_(True)
        sd = torch.ones(B, dtype=torch.float64) if scale is None else scale.double()
        n64 = torch.nn.functional.layer_norm(xr, (C,), gm, bt, 1e-5)
        pre = n64 @ w1.T + b1
        y64 = (_gelu64(pre) @ w2.T + b2) * sd[:, None, None] + xr
        y64.backward(dy.double().cpu())
        ref = [y64.detach(), n64.detach(), xr.grad] + [t.grad for t in (gm, bt, w1, b1, w2, b2)]
        res = {}
        for fused_ln in (True, False):
            ops.FFN_FUSED.ln = fused_ln
            for p in ps:
                p.grad.zero_()
            xx = x.clone().requires_grad_(True)
            ops.RANGES.begin(cuda)
            n0 = ops.FFN_FUSED.ln_calls
            n, xres = ops.layer_norm_fork(xx, ps[0], ps[1], lazy=True)
            y = ops.mlp(n, [(ps[2], ps[3]), (ps[4], ps[5])], act='gelu', identity=xres, out_scale=None if scale is None else scale.to(cuda))
            y.backward(dy)
            ops.flush_deferred()
            torch.cuda.synchronize()
            assert (ops.FFN_FUSED.ln_calls - n0 == 1) == fused_ln
            res[fused_ln] = [y.detach(), n.detach(), xx.grad.detach()] + [p.grad.detach().clone() for p in ps]
        errs = [(i, _rel(a, r), _rel(b, r)) for i, (a, b, r) in enumerate(zip(res[True], res[False], ref))]
        assert all(torch.isfinite(a).all() for a in res[True])
        assert all(e_f <= max(1e-6, 1.5 * e_u) for _, e_f, e_u in errs), errs
        # the norm's output range word: the true maximum, as the norm's own launch leaves it
    finally:
        ops.FFN_FUSED.ln = old
        ops.DEFER.drop()
        opt.close()


@pytest.mark.parametrize('B,L,K,N,norm,resid', [(2, 16384, 96, 288, True, False), (2, 16384, 96, 96, False, True), (2, 4096, 192, 576, True, False),
                                                (2, 4096, 192, 192, False, True), (1, 9000, 96, 288, True, False), (3, 3000, 192, 192, False, True),
                                                (2, 4096, 384, 192, False, False), (2, 4096, 96, 288, False, False),
                                                (2, 1024, 384, 1152, True, False), (2, 1024, 384, 384, False, True), (2, 256, 768, 2304, False, False),
                                                (2, 256, 768, 768, False, True), (3, 300, 384, 256, False, False)])
def test_tall_narrow_linear_on_the_rows_resident_launch(cuda, B, L, K, N, norm, resid):
    """One Linear of the Swin window attention at stage 1 / 2 sizes on ops.LIN_FUSED (rscotr_lin_h3 / _ln): y = [LayerNorm](x) W^T + b
    [* DropPath factor + identity] forward, the input gradient (the mirrored launch: K and N swapped, planes of W^T) and every parameter
    gradient against fp64 and against the tiled route, ragged row counts included."""
    from rscotr_amd import ops
    from rscotr_amd.optim import FlatAdamW
    if not ops.RANGES.enabled:
        pytest.skip('value ranges are off')
    g = torch.Generator().manual_seed(B * L + K + N)
    x = (torch.randn(B, L, K, generator=g) * 1.3 + 0.2).to(cuda)
    dy = torch.randn(B, L, N, generator=g).to(cuda)
    ident = torch.randn(B, L, N, generator=g).to(cuda) if resid else None
    scale = torch.tensor([1.25, 0.0, 1.25][:B]) if resid else None
    ps = [torch.nn.Parameter((torch.randn(sh, generator=g) * sc + off).to(cuda))
          for sh, sc, off in (((K,), 0.4, 1.0), ((K,), 0.3, 0.0), ((N, K), 0.1, 0.0), ((N,), 0.3, 0.0))]
    opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
    old, old_rows = ops.LIN_FUSED.enabled, ops.LIN_FUSED.MIN_ROWS
    ops.LIN_FUSED.MIN_ROWS = 8192 if K in (384, 768) and B * L < 4096 else 1024
    try:
        gm, bt, w, b = [p.detach().double().cpu().requires_grad_(True) for p in ps]
        xr = x.double().cpu().requires_grad_(True)
        sd = torch.ones(B, dtype=torch.float64) if scale is None else scale.double()
        n64 = torch.nn.functional.layer_norm(xr, (K,), gm, bt, 1e-5) if norm else xr
        y64 = (n64 @ w.T + b) * sd[:, None, None] + (ident.double().cpu() if resid else 0)
        y64.backward(dy.double().cpu())
        ref = [y64.detach(), xr.grad, w.grad, b.grad] + ([gm.grad, bt.grad] if norm else [])
        res = {}
        for fused in (True, False):
            ops.LIN_FUSED.enabled = fused
            for p in ps:
                p.grad.zero_()
            xx = x.clone().requires_grad_(True)
            ops.RANGES.begin(cuda)
            n0, l0 = ops.LIN_FUSED.calls, ops.LIN_FUSED.ln_calls
            h = ops.layer_norm_fork(xx, ps[0], ps[1], lazy=True)[0] if norm else xx
            y = ops.linear(h, ps[2], ps[3], resid=ident, out_scale=None if scale is None else scale.to(cuda))
            y.backward(dy)
            ops.flush_deferred()
            torch.cuda.synchronize()
            # (the input gradient of a FEW-row case runs over K' = N: on the launch only where that is one of its widths)
            back = 1 if (B * L >= 8192 or N in (384, 768)) else 0
            assert ops.LIN_FUSED.calls - n0 == ((1 + back) if fused else 0), ops.LIN_FUSED.calls - n0
            assert ops.LIN_FUSED.ln_calls - l0 == (1 if fused and norm else 0)
            res[fused] = [y.detach(), xx.grad.detach(), ps[2].grad.detach().clone(), ps[3].grad.detach().clone()] + \
                ([ps[0].grad.detach().clone(), ps[1].grad.detach().clone()] if norm else [])
        errs = [(i, _rel(a, r), _rel(bb, r)) for i, (a, bb, r) in enumerate(zip(res[True], res[False], ref))]
        assert all(torch.isfinite(a).all() for a in res[True])
        assert all(e_f <= max(1e-6, 1.5 * e_u) for _, e_f, e_u in errs), errs
    finally:
        ops.LIN_FUSED.enabled, ops.LIN_FUSED.MIN_ROWS = old, old_rows
        ops.DEFER.drop()
        opt.close()
