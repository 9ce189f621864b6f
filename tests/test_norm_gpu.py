"""LayerNorm variants of the product path against torch on the CPU in fp64 (tolerance 1e-3 relative, north star; observed
~1e-6):

* `ops.layer_norm_sum`: the norm and `norm + positional embedding` from one launch (mmcv BaseTransformerLayer `norm` followed
  by an attention wrapper's `query + query_pos`), the sum carrying no gradient;
* `ops.patch_merge_norm`: mmcv PatchMerging's nn.Unfold(2, stride 2) + LayerNorm(4 Cin) with the unfold done by the norm
  kernels' loads / stores, odd maps included (zero "corner" padding);
* attention nodes fed a producer-formed sum (`q_sum` / `k_sum`) return the same outputs and gradients as when they add the
  embeddings themselves."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.detach().double()
    return float((a.detach().cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize('B,L,C,expanded', [(2, 800, 256, False), (2, 100, 256, True), (3, 37, 64, False), (2, 5440, 256, False),
                                            (1, 9, 1024, True)])
def test_layer_norm_sum(cuda, B, L, C, expanded):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + L + C)
    x = torch.randn(B, L, C, generator=g) * 2 + 0.3
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    pos = torch.randn(1 if expanded else B, L, C, generator=g)
    dy = torch.randn(B, L, C, generator=g)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), wr, br, 1e-5)
    yr.backward(dy.double())
    xd = x.to(cuda).requires_grad_(True)
    wd, bd = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    pd = pos.to(cuda).requires_grad_(True)
    y, s = ops.layer_norm_sum(xd, wd, bd, pd.expand(B, -1, -1) if expanded else pd)
    assert not s.requires_grad and y.requires_grad
    y.backward(dy.to(cuda))
    torch.cuda.synchronize()
    assert _rel(y, yr) <= 1e-5
    assert _rel(s, yr.detach() + pos.double()) <= 1e-5
    assert _rel(xd.grad, xr.grad) <= 1e-4 and _rel(wd.grad, wr.grad) <= 1e-4 and _rel(bd.grad, br.grad) <= 1e-4
    assert pd.grad is None   # the sum is data: d(pos) comes from the attention that consumes it


@pytest.mark.parametrize('B,H,W,Cin', [(2, 128, 128, 96), (2, 64, 64, 192), (2, 32, 32, 384), (1, 7, 9, 96), (2, 5, 5, 128),
                                       (1, 14, 14, 512), (3, 6, 4, 8), (2, 9, 8, 256), (2, 6, 6, 6)])
def test_patch_merge_norm(cuda, B, H, W, Cin):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(H * 131 + W * 7 + Cin)
    x = torch.randn(B, H * W, Cin, generator=g) + 0.2
    C = 4 * Cin
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn(B, Ho * Wo, C, generator=g)
    # reference: mmcv PatchMerging = pad to even, nn.Unfold(2, stride 2) on the (B, Cin, H, W) map, LayerNorm(4 Cin)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    m = xr.view(B, H, W, Cin).permute(0, 3, 1, 2)
    m = F.pad(m, (0, W % 2, 0, H % 2))
    u = F.unfold(m, 2, stride=2).transpose(1, 2)            # (B, Ho*Wo, Cin*4), channel-major c*4 + kh*2 + kw
    yr = F.layer_norm(u, (C,), wr, br, 1e-5)
    yr.backward(dy.double())
    xd = x.to(cuda).requires_grad_(True)
    wd, bd = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    y, hw2 = ops.patch_merge_norm(xd, (H, W), wd, bd)
    assert hw2 == (Ho, Wo) and y.shape == (B, Ho * Wo, C)
    y.backward(dy.to(cuda))
    torch.cuda.synchronize()
    assert _rel(y, yr) <= 1e-5
    assert _rel(xd.grad, xr.grad) <= 1e-4 and _rel(wd.grad, wr.grad) <= 1e-4 and _rel(bd.grad, br.grad) <= 1e-4
    # and the gathered-copy route of the same op (what the fused launch replaces) gives the same numbers
    x2 = x.to(cuda).requires_grad_(True)
    y2 = ops.layer_norm(ops.patch_merge_gather(x2, (H, W))[0], wd.detach(), bd.detach())
    y2.backward(dy.to(cuda))
    assert torch.equal(y.detach(), y2.detach()) and torch.equal(xd.grad, x2.grad)   # same ownership and summation order


def test_attention_nodes_accept_producer_formed_sums(cuda):
    """q_sum / k_sum change where the sum is formed, not what is computed: outputs and every gradient equal the run in
    which the node adds the embeddings itself (bitwise: the same kernels on the same values)."""
    from rscotr_amd import ops
    from rscotr_amd.layers import LevelGeometry
    torch.manual_seed(5)
    B, Lq, Lk, C, heads = 2, 100, 256, 256, 8
    mha = torch.nn.MultiheadAttention(C, heads, 0.0).to(cuda)

    def run_mha(with_sums):
        x = torch.randn(B, Lq, C, device=cuda, generator=torch.Generator(cuda).manual_seed(1)).requires_grad_(True)
        qp = torch.randn(B, Lq, C, device=cuda, generator=torch.Generator(cuda).manual_seed(2)).requires_grad_(True)
        kx = torch.randn(B, Lk, C, device=cuda, generator=torch.Generator(cuda).manual_seed(3)).requires_grad_(True)
        kp = torch.randn(B, Lk, C, device=cuda, generator=torch.Generator(cuda).manual_seed(4))
        for p in mha.parameters():
            p.grad = None
        kw = dict(q_sum=(x + qp).detach(), k_sum=(kx + kp).detach()) if with_sums else {}
        y = ops.mha(x, kx, kx, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, heads,
                    identity=x, q_pos=qp, k_pos=kp, **kw)
        y.square().sum().backward()
        return [y.detach(), x.grad, qp.grad, kx.grad] + [p.grad.clone() for p in mha.parameters()]

    for a, b in zip(run_mha(False), run_mha(True)):
        assert torch.equal(a, b)

    shapes = [(16, 16), (8, 8), (4, 4), (2, 2)]
    geom = LevelGeometry.get(shapes, cuda)
    N = geom.num_tokens
    from rscotr_amd.layers import MultiScaleDeformableAttention
    m = MultiScaleDeformableAttention(C, heads, 4, 4, dropout=0.0).to(cuda)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.attention_weights.weight.normal_(0, 0.02)
    ref = torch.rand(B, N, 4, 2, device=cuda)

    def run_msda(with_sums):
        x = torch.randn(B, N, C, device=cuda, generator=torch.Generator(cuda).manual_seed(7)).requires_grad_(True)
        qp = torch.randn(B, N, C, device=cuda, generator=torch.Generator(cuda).manual_seed(8)).requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(x, None, None, None, query_pos=qp, reference_points=ref, query_sum=(x + qp).detach() if with_sums else None,
              **geom.kwargs())
        y.square().sum().backward()
        return [y.detach(), x.grad, qp.grad] + [p.grad.clone() for p in m.parameters()]

    for a, b in zip(run_msda(False), run_msda(True)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('M,dims', [(1600, (512, 256, 256)), (200, (512, 256, 256)), (37, (64, 128, 32))])
def test_mlp_sum_output(cuda, M, dims):
    """ops.mlp(..., sum_with=s) -> (y, y + s): y and its gradients as without the sum, the sum carries values only (the
    `query + query_pos` a DINO decoder layer's first attention would form from the positional MLP's output)."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M + dims[0])
    x = torch.randn(2, M // 2 if M % 2 == 0 else M, dims[0], generator=g)
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) * 0.1 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    s = torch.randn(*x.shape[:-1], dims[-1], generator=g)
    dy = torch.randn(*x.shape[:-1], dims[-1], generator=g)

    def run(with_sum):
        xd = x.to(cuda).requires_grad_(True)
        layers = [(w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)) for w, b in zip(Ws, bs)]
        sd = s.to(cuda).requires_grad_(True)
        out = ops.mlp(xd, layers, act='relu', sum_with=sd if with_sum else None)
        y, ysum = out if with_sum else (out, None)
        y.backward(dy.to(cuda))
        torch.cuda.synchronize()
        assert sd.grad is None
        return y.detach(), ysum, xd.grad, [w.grad for w, _ in layers], [b.grad for _, b in layers], sd

    y0, _, gx0, gw0, gb0, _ = run(False)
    y1, ysum, gx1, gw1, gb1, sd = run(True)
    assert not ysum.requires_grad
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    for a, b in zip(gw0 + gb0, gw1 + gb1):
        assert torch.equal(a, b)
    assert torch.allclose(ysum, y1 + sd.detach(), rtol=0, atol=1e-6)
    ref = x.double()
    for i, (w, b) in enumerate(zip(Ws, bs)):
        ref = ref @ w.double().t() + b.double()
        if i < len(Ws) - 1:
            ref = torch.relu(ref)
    assert _rel(y1, ref) <= 1e-5
