"""Weight operands of the fp16 split product from pre-split planes (rscotr_gemm_split_weights_h3 / rscotr_gemm_f32_rb,
ops.HPLANES): bit-identical to the in-kernel split — same planes — for y = x W^T and dx = dy W, for slices of a parameter, and
after the optimizer has changed the weights (the sets are re-split)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _products(ops, x, dy, w, wslice):
    """forward y = x W^T (+ bias, ReLU), dx = dy W through act', and both once more on a row slice of the parameter"""
    M, K = x.shape
    N = w.shape[0]
    b = torch.linspace(-1, 1, N, device=x.device)
    y = ops.gemm(x, w, M, N, K, K, K, 0, 0, bias=b, act=ops.ACT_RELU)
    dx = ops.gemm(dy, w, M, K, N, N, K, 0, 1, resid=x)
    ys = ops.gemm(x, wslice, M, wslice.shape[0], K, K, K, 0, 0)
    return y, dx, ys


@pytest.mark.parametrize('M,N,K', [(16384, 256, 256), (10880, 256, 2048), (8192, 768, 192)])
def test_planes_route_is_bit_identical_and_follows_the_optimizer(cuda, M, N, K):
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    from rscotr_amd.optim import FlatAdamW
    if not ops.RANGES.enabled:
        pytest.skip('the plane operands belong to the fp16 split product')
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(N, K, device=cuda) * 0.05)
    big = torch.nn.Parameter(torch.randn(3 * 256, K, device=cuda) * 0.1)  # (an in_proj-like parameter used by row slices)
    opt = FlatAdamW([dict(name='w', param=w, lr=1e-2, weight_decay=0.05), dict(name='big', param=big, lr=1e-2, weight_decay=0.0)],
                    grad_clip=dict(max_norm=0.1))
    old = ops.HPLANES.enabled
    try:
        x = torch.randn(M, K, device=cuda)
        dy = torch.randn(M, N, device=cuda)
        assert lib.rscotr_gemm_f32_split_route(M, N, K, K, K, 0, 0, ops.ACT_RELU, 0, 0, 0, 0) == 2
        for step in range(3):
            ops.RANGES.begin(cuda)
            ops.HPLANES.enabled = False
            ref = _products(ops, x, dy, w.data, big.data[256:512])
            ops.RANGES.begin(cuda)
            ops.HPLANES.enabled = True
            n0 = len(ops.HPLANES.entries)
            got = _products(ops, x, dy, w.data, big.data[256:512])
            assert step or len(ops.HPLANES.entries) >= n0 + 2, 'the plane route was not taken'  # (forward + slice; dx where its shape routes to the 64 x 64 kernel)
            for a, r in zip(got, ref):
                assert torch.equal(a, r)
            # the weights change: the next products must see new planes
            opt.zero_grad()
            w.grad.copy_(torch.randn_like(w))
            big.grad.copy_(torch.randn_like(big))
            opt._on_ready(0)
            opt._on_ready(1)
            opt.step()
    finally:
        ops.HPLANES.enabled = old
        opt.close()
