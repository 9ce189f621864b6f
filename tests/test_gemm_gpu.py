"""Parity of the fp32 MFMA GEMM / fused MLP / LayerNorm kernels (through the C ABI) against fp64
CPU references of the same contractions.  Tolerance: 1e-3 relative (north star); the observed
error of an fp32 fmaf chain is ~1e-6."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.double()
    return float((a.detach().cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (3, 45, 48), (130, 96, 96), (257, 288, 100), (64, 64, 256),
                                   (200, 20, 256), (500, 128, 2048), (1000, 768, 3072), (50, 2048, 256),
                                   (33, 31, 19), (4096, 32, 64),
                                   # the low-latency small-product kernel (K % 8 == 0, K <= 512, <= 1024 tiles of 32x32)
                                   (200, 256, 256), (1600, 256, 256), (256, 256, 200), (1600, 4, 256), (37, 45, 32),
                                   (31, 33, 40), (100, 64, 512), (2048, 384, 384)])
@pytest.mark.parametrize('ak,bk', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_layouts(cuda, M, N, K, ak, bk):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ak * 2 + bk)
    A = torch.randn((K, M) if ak else (M, K), generator=g)
    B = torch.randn((K, N) if bk else (N, K), generator=g)
    ref = (A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t())
    out = ops.gemm(A.to(cuda), B.to(cuda), M, N, K, A.shape[1], B.shape[1], ak, bk)
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize('act', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('M,N,K', [(300, 200, 128), (300, 200, 1024), (130, 77, 256)])  # small-product and tiled kernels
def test_gemm_epilogues(cuda, act, M, N, K):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(act)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, aux, resid, c0 = (torch.randn(N, generator=g), torch.randn(M, N, generator=g),
                            torch.randn(M, N, generator=g), torch.randn(M, N, generator=g))
    v = A.double() @ B.double().t() + bias.double()
    pre_ref = v.clone()
    if act == 1:
        v = v.clamp(min=0)
    elif act == 2:
        v = F.gelu(v)
    elif act == 3:
        v = v * (aux > 0)
    elif act == 4:
        a = aux.double().requires_grad_(True)
        F.gelu(a).sum().backward()
        v = v * a.grad
    ref = v + resid.double() + c0.double()
    out = c0.clone().to(cuda)
    pre = torch.empty(M, N, device=cuda)
    ops.gemm(A.to(cuda), B.to(cuda), M, N, K, K, K, 0, 0, out=out, bias=bias.to(cuda), act=act,
             aux=aux.to(cuda), pre=pre, resid=resid.to(cuda), accumulate=True)
    assert _rel(out, ref) < 1e-5
    assert _rel(pre, pre_ref) < 1e-5

@pytest.mark.parametrize('M,N,K,bk', [(200, 256, 256, 1), (1600, 256, 512, 1), (10880, 256, 384, 1), (10880, 256, 256, 0),
                                      (400, 256, 2048, 1), (130, 77, 100, 1), (2048, 384, 1536, 1)])
@pytest.mark.parametrize('acc', [False, True])
@pytest.mark.parametrize('with_resid', [True, False])
def test_gemm_second_output(cuda, M, N, K, bk, acc, with_resid):
    """out2 of rscotr_gemm_f32: C = A B (+ old C) stays WITHOUT the residual, C2 = C + resid — on the small-product,
    tiled, split-K and bf16x6 kernels (the merged input gradients of the attention blocks)."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn((K, N) if bk else (N, K), generator=g) * 0.1
    resid, c0 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ref = A.double() @ (B.double() if bk else B.double().t()) + (c0.double() if acc else 0.0)
    out = c0.clone().to(cuda) if acc else None
    out2 = torch.full((M, N), float('nan'), device=cuda)
    out = ops.gemm(A.to(cuda), B.to(cuda), M, N, K, K, B.shape[1], 0, bk, out=out, accumulate=acc,
                   resid=resid.to(cuda) if with_resid else None, out2=out2)
    assert _rel(out, ref) < 1e-5
    assert _rel(out2, ref + (resid.double() if with_resid else 0.0)) < 1e-5

def _split_planes(lib, W, rows, red, ldw, tr, cuda):
    """planes of rscotr_gemm_split_weights for one weight: rows = output rows of the plane set, red = reduction length."""
    import numpy as np
    npad = (rows + 255) // 256 * 256
    planes = torch.full((npad * red * 3,), 0x7fc0, dtype=torch.int16, device=cuda)  # bf16 NaN pattern: every word must be written
    blocks = (npad * (red // 16) + 255) // 256
    n_w, k_w = (red, rows) if tr else (rows, red)
    table = torch.from_numpy(np.asarray([[W.data_ptr(), planes.data_ptr(), n_w, k_w, ldw, npad, 0, tr]], dtype=np.int64)).to(cuda)
    lib.call('rscotr_gemm_split_weights', table.data_ptr(), 1, blocks, torch.cuda.current_stream().cuda_stream)
    return planes, npad


@pytest.mark.parametrize('M,N,K', [(10880, 256, 2048), (10880, 2048, 256), (2048, 384, 1536), (8192, 288, 192), (1600, 256, 256),
                                   (1000, 200, 112), (300, 45, 64), (13294, 256, 272), (2500, 3072, 768), (32768, 96, 384)])
@pytest.mark.parametrize('tr', [0, 1])
def test_gemm_with_presplit_weight_planes(cuda, six_term, M, N, K, tr):
    _presplit_weight_planes(cuda, M, N, K, tr)


def _presplit_weight_planes(cuda, M, N, K, tr):
    """rscotr_gemm_split_weights + rscotr_gemm_f32_wplanes against fp64: both plane orientations (y = x W^T with W (N, K);
    dx = dy W with W (K, N)), ragged M / N, k-slices, bias + activation / residual / second output epilogues; error of the
    class of an fp32 FMA chain (same bound as the in-kernel split)."""
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + tr)
    A = torch.randn(M, K, generator=g)
    W = torch.randn((K, N) if tr else (N, K), generator=g) * 0.1
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Wd = A.to(cuda), W.to(cuda)
    planes, npad = _split_planes(lib, Wd, N, K, W.shape[1], tr, cuda)
    ref = A.double() @ (W.double() if tr else W.double().t())
    s = torch.cuda.current_stream().cuda_stream
    nws = lib.rscotr_gemm_f32_wplanes_workspace(M, N, K)
    ws = torch.empty(max(nws, 4) // 4, device=cuda)
    out = torch.full((M, N), float('nan'), device=cuda)
    lib.call('rscotr_gemm_f32_wplanes', Ad.data_ptr(), planes.data_ptr(), npad, out.data_ptr(), M, N, K, K, N, 0, 0, 0, 0, 0, 0,
             0, 0, 0, ws.data_ptr(), nws, s)
    err = _rel(out, ref)
    fp32 = _rel(ops.gemm(Ad, Wd, M, N, K, K, W.shape[1], 0, tr), ref)
    assert err <= max(1e-6, 1.5 * fp32), (err, fp32)
    # epilogue: bias + ReLU, then + resid as the second output
    out2 = torch.full((M, N), float('nan'), device=cuda)
    bias_d, resid_d = bias.to(cuda), resid.to(cuda)  # (kept alive across the launch)
    lib.call('rscotr_gemm_f32_wplanes', Ad.data_ptr(), planes.data_ptr(), npad, out.data_ptr(), M, N, K, K, N,
             bias_d.data_ptr(), 1, 0, 0, resid_d.data_ptr(), 0, 0, 0, out2.data_ptr(), ws.data_ptr(), nws, s)
    want = torch.relu(ref + bias.double())
    assert _rel(out, want) < 2e-6 and _rel(out2, want + resid.double()) < 2e-6


def test_gemm_splitk_matches_unsplit(cuda):
    """dW-shaped problem (small output, long reduction) takes the split-K path."""
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    g = torch.Generator().manual_seed(5)
    M, N, K = 384, 96, 32768
    assert lib.rscotr_gemm_f32_workspace(M, N, K) > 0
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    ref = A.double().t() @ B.double()
    out = ops.gemm(A.to(cuda), B.to(cuda), M, N, K, M, N, 1, 1)
    assert _rel(out, ref) < 1e-5


@pytest.mark.parametrize('M,N,K', [(384, 96, 32768), (256, 256, 10880), (2048, 256, 1580), (45, 768, 2), (256, 20, 1580),
                                   (130, 70, 4100), (4, 256, 1580), (64, 64, 31), (256, 256, 200), (256, 20, 256),
                                   (45, 300, 512), (2048, 256, 200)])
def test_gemm_rowsum_rides_dw(cuda, M, N, K):
    """dW contraction with the bias gradient (row sums of the k-major A) from the same launch, split and
    unsplit, overwrite and accumulate; repeated calls give bit-identical results."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    ref = A.double().t() @ B.double()
    rs_ref = A.double().sum(0)
    Ad, Bd = A.to(cuda), B.to(cuda)
    outs = []
    for rep in range(3):
        rs = torch.full((M,), 7.0, device=cuda)
        out = ops.gemm(Ad, Bd, M, N, K, M, N, 1, 1, rowsum=rs)
        assert _rel(out, ref) < 1e-5
        assert _rel(rs, rs_ref) < 1e-5
        outs.append((out.clone(), rs.clone()))
    # deterministic: the slabs are combined in a fixed order whichever workgroup finishes the tile
    assert all(torch.equal(o, outs[0][0]) and torch.equal(r, outs[0][1]) for o, r in outs[1:])
    base = torch.randn(M, generator=g)
    c0 = torch.randn(M, N, generator=g)
    rs = base.clone().to(cuda)
    out = ops.gemm(Ad, Bd, M, N, K, M, N, 1, 1, out=c0.clone().to(cuda), accumulate=True, rowsum=rs,
                   rowsum_accumulate=True)
    assert _rel(out, ref + c0.double()) < 1e-5
    assert _rel(rs, rs_ref + base.double()) < 1e-5


def test_gemm_rowsum_needs_kmajor_a(cuda):
    from rscotr_amd import ops
    A, B = torch.randn(8, 4, device=cuda), torch.randn(8, 4, device=cuda)
    with pytest.raises(RuntimeError):
        ops.gemm(A, B, 8, 8, 4, 4, 4, 0, 0, rowsum=torch.zeros(8, device=cuda))


def test_colsum(cuda):
    from rscotr_amd import ops
    for M, N in [(1, 5), (700, 45), (10880, 256), (3, 2048)]:
        X = torch.randn(M, N)
        assert _rel(ops.colsum(X.to(cuda), M, N), X.double().sum(0)) < 1e-5


@pytest.mark.parametrize('act', ['relu', 'gelu'])
@pytest.mark.parametrize('nl,ident', [(1, False), (2, True), (3, False), (2, 'other')])
def test_mlp_autograd(cuda, act, nl, ident):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(11)
    dims = [96, 384, 96] if nl == 2 else ([96, 45] if nl == 1 else [256, 256, 256, 4])
    x = torch.randn(2, 77, dims[0], generator=g)
    layers = [(torch.randn(dims[i + 1], dims[i], generator=g) * 0.1, torch.randn(dims[i + 1], generator=g))
              for i in range(nl)]
    other = torch.randn(2, 77, dims[-1], generator=g)
    go = torch.randn(2, 77, dims[-1], generator=g)

    def ref():
        xr = x.double().requires_grad_(True)
        o = other.double().requires_grad_(True)
        ls = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in layers]
        h = xr
        for i, (w, b) in enumerate(ls):
            h = F.linear(h, w, b)
            if i < nl - 1:
                h = F.relu(h) if act == 'relu' else F.gelu(h)
        if ident is True:
            h = h + xr
        elif ident == 'other':
            h = h + o
        (h * go.double()).sum().backward()
        return h, xr.grad, o.grad, [(w.grad, b.grad) for w, b in ls]

    y_ref, dx_ref, do_ref, dl_ref = ref()
    xd = x.to(cuda).requires_grad_(True)
    od = other.to(cuda).requires_grad_(True)
    ld = [(w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)) for w, b in layers]
    y = ops.mlp(xd, ld, act=act, identity=xd if ident is True else (od if ident == 'other' else None))
    (y * go.to(cuda)).sum().backward()
    assert _rel(y, y_ref) < 1e-5
    assert _rel(xd.grad, dx_ref) < 1e-5
    if ident == 'other':
        assert _rel(od.grad, do_ref) < 1e-5
    for (w, b), (dw, db) in zip(ld, dl_ref):
        assert _rel(w.grad, dw) < 1e-5
        assert _rel(b.grad, db) < 1e-5


@pytest.mark.parametrize('act,nl,L', [('gelu', 2, 77), ('gelu', 2, 128), ('relu', 1, 64), ('relu', 3, 50)])
def test_mlp_out_scale_droppath(cuda, act, nl, L):
    """x + s_b * MLP(x) with the per-sample factor folded into the last Linear (epilogue row scale forward and
    in dH, operand scaling in dW / db) against the plain formulation in fp64; samples with s = 0 included."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(nl * 10 + L)
    B = 3
    dims = [96, 384, 96] if nl == 2 else ([128, 128] if nl == 1 else [64, 256, 256, 64])
    x = torch.randn(B, L, dims[0], generator=g)
    layers = [(torch.randn(dims[i + 1], dims[i], generator=g) * 0.1, torch.randn(dims[i + 1], generator=g))
              for i in range(nl)]
    scale = torch.tensor([1.25, 0.0, 1.25])
    go = torch.randn(B, L, dims[-1], generator=g)
    xr = x.double().requires_grad_(True)
    ls = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in layers]
    h = xr
    for i, (w, b) in enumerate(ls):
        h = F.linear(h, w, b)
        if i < nl - 1:
            h = F.relu(h) if act == 'relu' else F.gelu(h)
    y_ref = xr + h * scale.double().view(B, 1, 1)
    (y_ref * go.double()).sum().backward()
    xd = x.to(cuda).requires_grad_(True)
    ld = [(w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)) for w, b in layers]
    y = ops.mlp(xd, ld, act=act, identity=xd, out_scale=scale.to(cuda))
    (y * go.to(cuda)).sum().backward()
    assert _rel(y, y_ref) < 1e-5 and _rel(xd.grad, xr.grad) < 1e-5
    for (w, b), (wr, br) in zip(ld, ls):
        assert _rel(w.grad, wr.grad) < 1e-5 and _rel(b.grad, br.grad) < 1e-5


@pytest.mark.parametrize('M,C', [(1, 96), (100, 96), (333, 192), (70, 256), (129, 384), (65, 768), (40, 1536),
                                 (16384, 96), (7, 2048), (9, 32)])
def test_layernorm(cuda, M, C):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 3 + 1
    w, b, go = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(M, C, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    (yr * go.double()).sum().backward()
    xd, wd, bd = (t.to(cuda).requires_grad_(True) for t in (x, w, b))
    y = ops.layer_norm(xd, wd, bd)
    (y * go.to(cuda)).sum().backward()
    assert _rel(y, yr) < 1e-5
    assert _rel(xd.grad, xr.grad) < 1e-4
    assert _rel(wd.grad, wr.grad) < 1e-4
    assert _rel(bd.grad, br.grad) < 1e-4


@pytest.mark.parametrize('M,C', [(100, 96), (333, 192), (4097, 384), (65, 768)])
def test_layernorm_fork(cuda, M, C):
    """(LayerNorm(x), x) of a pre-norm residual block: the residual gradient is added inside the LayerNorm backward
    kernel (dx_add of rscotr_layernorm_bwd) — same x.grad as the two-branch graph in fp64."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M * 3 + C)
    x = torch.randn(M, C, generator=g) * 2 - 0.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    go, gr = torch.randn(M, C, generator=g), torch.randn(M, C, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    ((yr * go.double()).sum() + (xr * gr.double()).sum()).backward()
    xd, wd, bd = (t.to(cuda).requires_grad_(True) for t in (x, w, b))
    y, xres = ops.layer_norm_fork(xd, wd, bd)
    assert torch.equal(xres, xd)
    ((y * go.to(cuda)).sum() + (xres * gr.to(cuda)).sum()).backward()
    assert _rel(y, yr) < 1e-5
    assert _rel(xd.grad, xr.grad) < 1e-4
    assert _rel(wd.grad, wr.grad) < 1e-4 and _rel(bd.grad, br.grad) < 1e-4
    # only the residual output used: the gradient passes through untouched
    x2 = x.to(cuda).requires_grad_(True)
    _, xres = ops.layer_norm_fork(x2, wd, bd)
    (xres * gr.to(cuda)).sum().backward()
    assert torch.equal(x2.grad, gr.to(cuda))


@pytest.mark.parametrize('M,N,K,ak,bk', [(10880, 2048, 256, 0, 0), (10880, 256, 2048, 0, 1), (2048, 1536, 384, 0, 0),
                                         (8192, 192, 768, 0, 1), (2048, 1024, 96, 1, 0), (256, 2048, 10880, 1, 1),
                                         (384, 1536, 2048, 1, 1), (256, 256, 10880, 1, 1), (4096, 4096, 4096, 0, 0),
                                         (1000, 768, 3072, 0, 0),
                                         # ragged shapes (EDGE instantiations: clamped loads, zeros past K, guarded stores): the
                                         # 800 x 800 det step's 4 x 13 294 rows (configs[3]) as M and as the reduction of a
                                         # weight gradient, Swin stage 1's N = 96, ragged rows of k-major operands
                                         (53176, 256, 256, 0, 0), (53176, 256, 256, 0, 1), (256, 2048, 53176, 1, 1),
                                         (32768, 96, 384, 0, 0), (2000, 1024, 512, 1, 0), (8192, 200, 512, 0, 1),
                                         (1604, 2048, 256, 0, 0)])
def test_gemm_bf16x6_is_fp32_accurate(cuda, gemm_precision, six_term, M, N, K, ak, bk):
    """Precision mode 3 (gemm_bf16x6_kernel: three bf16 planes per fp32 operand, six MFMAs per k-step, fp32 accumulate)
    against fp64 next to the fp32 matrix pipe on the step's own shapes, all four operand layouts, both tile sizes, with
    the fused epilogue: its error must be of the fp32 FMA chain's class — at most 1e-6 of max|C| (or 1.5x the fp32 pipe's own error) up
    to K = 2048, never more than twice the fp32 pipe's + 5e-7 — where round 1's two-plane product has
    4-6e-6."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M + N + K + ak + bk)
    A = torch.randn((K, M) if ak else (M, K), generator=g)
    B = torch.randn((K, N) if bk else (N, K), generator=g) * 0.05
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = ((A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t()) + bias.double()).clamp(min=0) \
        + resid.double()
    err, outs = {}, {}
    for mode in (0, 3):
        gemm_precision(mode)
        outs[mode] = ops.gemm(A.to(cuda), B.to(cuda), M, N, K, A.shape[1], B.shape[1], ak, bk, bias=bias.to(cuda), act=1,
                              resid=resid.to(cuda))
        err[mode] = _rel(outs[mode], ref)
    # the shapes the dispatch rules of csrc/gemm.hip (choose_split6) send to the split product must really take it
    routed = (M, N, K, ak, bk) in {(10880, 2048, 256, 0, 0), (10880, 256, 2048, 0, 1), (2048, 1536, 384, 0, 0), (8192, 192, 768, 0, 1),
                                   (256, 2048, 10880, 1, 1), (384, 1536, 2048, 1, 1), (4096, 4096, 4096, 0, 0),
                                   (1000, 768, 3072, 0, 0), (53176, 256, 256, 0, 0), (53176, 256, 256, 0, 1),
                                   (256, 2048, 53176, 1, 1), (32768, 96, 384, 0, 0), (2000, 1024, 512, 1, 0),
                                   (8192, 200, 512, 0, 1), (1604, 2048, 256, 0, 0)}
    assert torch.equal(outs[0], outs[3]) != routed, (M, N, K, ak, bk, err)
    assert err[3] <= 2.0 * err[0] + 5e-7, err
    if K <= 2048:  # (the fp32 FMA chain itself reaches 9e-7 at K = 2048)
        assert err[3] <= max(1e-6, 1.5 * err[0]), err
    if M % 64 or N % 64:  # ragged: nothing written past the edge (the result tensor is exactly M x N: a stray store would
        assert torch.isfinite(outs[3]).all()  # have hit the allocator's neighbour; checked with a guard band below)
        guard = torch.full((M + 8, N), 7.0, device=cuda)
        ops.gemm(A.to(cuda), B.to(cuda), M, N, K, A.shape[1], B.shape[1], ak, bk, out=guard[:M], bias=bias.to(cuda), act=1,
                 resid=resid.to(cuda))
        assert torch.equal(guard[:M], outs[3]) and bool((guard[M:] == 7.0).all())


def test_gemm_bf16x6_weight_gradient_routes(cuda, gemm_precision, six_term):
    """The dW route of mode 3: k-slices through slabs (immediate combine and accumulate into C), the bias gradient riding
    along (row sums of the k-major A operand, taken from the fp32 registers) and per-sample k scaling (stochastic depth)."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 384, 1536, 2048
    G, X, ks = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g), torch.rand(2, generator=g) + 0.5
    Gs = G.double() * ks.double().repeat_interleave(K // 2)[:, None]
    ref, ref_b = Gs.t() @ X.double(), Gs.sum(0)
    C0 = torch.randn(M, N, generator=g)
    for mode in (0, 3):
        gemm_precision(mode)
        db = torch.empty(M, device=cuda)
        out = ops.gemm(G.to(cuda), X.to(cuda), M, N, K, M, N, 1, 1, rowsum=db, kscale=ks.to(cuda), krows_per=K // 2)
        assert _rel(out, ref) < 2e-6 and _rel(db, ref_b) < 1e-5, mode
        acc = C0.clone().to(cuda)
        ops.gemm(G.to(cuda), X.to(cuda), M, N, K, M, N, 1, 1, out=acc, accumulate=True, kscale=ks.to(cuda), krows_per=K // 2)
        assert _rel(acc, ref + C0.double()) < 2e-6, mode


@pytest.mark.parametrize('x6', [0, 1])
def test_grouped_deferred_weight_gradients(cuda, x6, monkeypatch, six_term):
    """(x6 = 0: every member on the fp32 pipe's 64 x 64 tiles; 1, the default: members with min(M, N) >= 48 on the split product's
    128 x 128 edge body.)  rscotr_gemm_dw_group through ops.DEFER: a batch of dW = A^T B problems of the step's small-output shapes (ragged
    M / N / K, a destination shared by two problems, bias gradients riding along, per-sample k scaling) computed by ONE
    grouped launch + the deferred combine, against fp64; destinations are ACCUMULATED into."""
    from rscotr_amd import ops
    monkeypatch.setattr(ops.DEFER, 'group_x6', x6)
    g = torch.Generator().manual_seed(21)
    shapes = [(256, 256, 10880), (256, 256, 1600), (200, 256, 256), (4, 256, 1600), (96, 48, 4096), (384, 384, 2048),
              (128, 256, 10880), (256, 256, 1600), (192, 192, 8192), (100, 20, 40)]
    assert not ops.DEFER.pending()
    keep, want, outs = [], [], []
    shared = None
    for i, (M, N, K) in enumerate(shapes):
        A = torch.randn(K, M, generator=g).to(cuda)
        B = (torch.randn(K, N, generator=g) * 0.05).to(cuda)
        ks = (torch.rand(2, generator=g) + 0.5).to(cuda) if i in (2, 5) and K % 2 == 0 else None
        if i == 7:  # second contraction into the destination of problem 1
            out, rs = shared
        else:
            out = torch.randn(M, N, generator=g).to(cuda)
            rs = torch.randn(M, generator=g).to(cuda) if i % 2 == 0 else None
            want.append([out.double().cpu(), None if rs is None else rs.double().cpu()])
            outs.append((out, rs))
            if i == 1:
                shared = (out, rs)
        Ad = A.double().cpu() * (1.0 if ks is None else ks.double().cpu().repeat_interleave(K // 2)[:, None])
        tgt = want[1] if i == 7 else want[-1]
        tgt[0] = tgt[0] + Ad.t() @ B.double().cpu()
        if rs is not None:
            tgt[1] = tgt[1] + Ad.sum(0)
        ops.DEFER.group.append((A.data_ptr(), B.data_ptr(), out.data_ptr(), 0 if rs is None else rs.data_ptr(),
                                0 if ks is None else ks.data_ptr(), M, N, K, M, N, K // 2 if ks is not None else 1))
        ops.DEFER.group_keep.extend(t for t in (A, B, ks) if t is not None)
        keep.append((A, B, ks))
    ops.flush_deferred()
    torch.cuda.synchronize()
    assert not ops.DEFER.pending()
    for (out, rs), (w, wr) in zip(outs, want):
        assert _rel(out, w) < 3e-6, (tuple(out.shape), _rel(out, w))
        if rs is not None:
            assert _rel(rs, wr) < 1e-5, tuple(out.shape)
