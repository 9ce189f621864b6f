"""The ReLU gate of a wide FFN as one bit per element (rscotr_gemm_relu_bits_ok, act 5 / 6 of rscotr_gemm_f32_r): the forward
product leaves [h > 0] as words, the gated backward product reads them instead of the M x N activation — bit-identical to
act 1 / act 3 (mmcv FFN: Linear -> ReLU -> Linear, cfg ...potsdam.py:86-93), and refused on products outside the interior
128 x 128 split-product tiles."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _slot(ops, lib, x):
    s = ops.RANGES.new_slot(x.device)
    lib.call('rscotr_amax_f32', x.data_ptr(), x.shape[0], x.shape[1], x.shape[1], s, torch.cuda.current_stream().cuda_stream)
    return s


@pytest.mark.parametrize('M,N,K', [(10880, 2048, 256), (8192, 1024, 256), (4096, 2048, 512)])
@pytest.mark.parametrize('ranges', [True, False])
def test_gate_bits_equal_the_activation_gate(cuda, gemm_precision, M, N, K, ranges):
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    assert lib.rscotr_gemm_relu_bits_ok(M, N, K, K, K, 0, 0) == 1 and lib.rscotr_gemm_relu_bits_ok(M, N, K, K, N, 0, 1) == 1
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g).to(cuda)
    W1 = (torch.randn((N, K), generator=g) * 0.05).to(cuda)
    b1 = torch.randn(N, generator=g).to(cuda)
    dy = torch.randn((M, K), generator=g).to(cuda)       # gradient of a second Linear N -> K
    W2 = (torch.randn((K, N), generator=g) * 0.05).to(cuda)
    kw = lambda a, b: dict(amax_a=_slot(ops, lib, a), amax_b=_slot(ops, lib, b)) if ranges else {}
    old = ops.RANGES.enabled
    ops.RANGES.enabled = ranges  # (False: nobody measures the operands -> the six-term bf16 product on the same tiles)
    try:
        _compare(ops, cuda, x, W1, b1, dy, W2, M, N, K, kw)
    finally:
        ops.RANGES.enabled = old


def _compare(ops, cuda, x, W1, b1, dy, W2, M, N, K, kw):
    # forward: act 1 against act 5
    h_ref = ops.gemm(x, W1, M, N, K, K, K, 0, 0, bias=b1, act=ops.core.ACT_RELU, **kw(x, W1))
    bits = torch.zeros(M * N // 64, dtype=torch.int64, device=cuda)
    h = ops.gemm(x, W1, M, N, K, K, K, 0, 0, bias=b1, act=ops.core.ACT_RELU_BITS, pre=bits, **kw(x, W1))
    assert torch.equal(h, h_ref)
    # every element's gate is in exactly one bit
    assert int((h_ref > 0).sum()) == sum(int(((bits >> s) & 1).sum()) for s in range(64))
    # backward: act 3 (reads h) against act 6 (reads the words)
    d_ref = ops.gemm(dy, W2, M, N, K, K, N, 0, 1, act=ops.core.ACT_RELU_GRAD, aux=h_ref, **kw(dy, W2))
    d = ops.gemm(dy, W2, M, N, K, K, N, 0, 1, act=ops.core.ACT_RELU_GRAD_BITS, aux=bits, **kw(dy, W2))
    assert torch.equal(d, d_ref)
    assert float(d.abs().max()) > 0


def test_gate_bits_are_refused_off_the_128_tiles(cuda, gemm_precision):
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    for M, N, K in [(200, 2048, 256), (10880, 2048, 200), (1600, 256, 256), (10880 + 64, 2048, 256)]:
        assert lib.rscotr_gemm_relu_bits_ok(M, N, K, K, K, 0, 0) == 0
    M, N, K = 1600, 256, 256
    x, W = torch.randn(M, K, device=cuda), torch.randn(N, K, device=cuda)
    bits = torch.zeros(M * N // 64, dtype=torch.int64, device=cuda)
    with pytest.raises(RuntimeError, match='rscotr_gemm_relu_bits_ok'):
        ops.gemm(x, W, M, N, K, K, K, 0, 0, act=ops.core.ACT_RELU_BITS, pre=bits)
    # with the codes: bias only
    M, N, K = 8192, 1024, 256
    x, W = torch.randn(M, K, device=cuda), torch.randn(N, K, device=cuda)
    bits = torch.zeros(M * N // 64, dtype=torch.int64, device=cuda)
    with pytest.raises(RuntimeError, match='bias only'):
        ops.gemm(x, W, M, N, K, K, K, 0, 0, act=ops.core.ACT_RELU_BITS, pre=bits, resid=torch.zeros(M, N, device=cuda))


def test_ffn_node_takes_the_bits_and_matches_the_plain_route(cuda, gemm_precision, monkeypatch):
    """ops.mlp on the encoder FFN shape (10880 x 256 -> 2048 -> 256): same output and gradients, bit for bit, with the gate as
    bits (the default) and with RSCOTR_RELU_BITS off; the node saves the words instead of nothing extra."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(7)
    M, C, F = 10880, 256, 2048
    x0 = torch.randn((2, M // 2, C), generator=g).to(cuda)
    W1, b1 = (torch.randn((F, C), generator=g) * 0.05).to(cuda), torch.randn(F, generator=g).to(cuda)
    W2, b2 = (torch.randn((C, F), generator=g) * 0.02).to(cuda), torch.randn(C, generator=g).to(cuda)
    dy = torch.randn((2, M // 2, C), generator=g).to(cuda)
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops.RELU_BITS, 'enabled', on)
        assert ops.RELU_BITS.ok(M, F, C, C) == (on and ops.RANGES.enabled)
        x = x0.clone().requires_grad_(True)
        ps = [t.clone().requires_grad_(True) for t in (W1, b1, W2, b2)]
        y = ops.mlp(x, [(ps[0], ps[1]), (ps[2], ps[3])], act='relu', identity=x)
        y.backward(dy)
        res.append([y.detach()] + [t.grad for t in [x] + ps])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_block_outputs_commit_no_range_word_and_values_do_not_change(cuda, gemm_precision, monkeypatch):
    """ops.RANGE_OUT: a product whose output no later product multiplies with (a Linear that takes a residual, a projection read by
    an attention kernel: ops.linear(range_out=False)) leaves no range word — one atomic round trip per workgroup less — and the
    numbers are the same as with every product committing (RSCOTR_RANGE_OUT_ALL=1)."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(11)
    M, C, F = 2048, 256, 1024
    x0 = torch.randn((2, M // 2, C), generator=g).to(cuda)
    W1, b1 = (torch.randn((F, C), generator=g) * 0.05).to(cuda), torch.randn(F, generator=g).to(cuda)
    W2, b2 = (torch.randn((C, F), generator=g) * 0.02).to(cuda), torch.randn(C, generator=g).to(cuda)
    dy = torch.randn((2, M // 2, C), generator=g).to(cuda)
    res = []
    for all_ in (False, True):
        monkeypatch.setattr(ops.RANGE_OUT, 'all', all_)
        x = x0.clone().requires_grad_(True)
        ps = [t.clone().requires_grad_(True) for t in (W1, b1, W2, b2)]
        y = ops.mlp(x, [(ps[0], ps[1]), (ps[2], ps[3])], act='relu', identity=x)
        q = ops.linear(x, ps[0], ps[1], range_out=False)
        k = ops.linear(x, ps[0], ps[1])
        if ops.RANGES.enabled:
            assert bool(ops.RANGES.slot_of(y)) == all_ and bool(ops.RANGES.slot_of(q)) == all_ and bool(ops.RANGES.slot_of(k))
        (y.sum() * 0 + (y * dy).sum() + q.sum() + k.sum()).backward()
        res.append([y.detach(), q.detach(), k.detach()] + [t.grad for t in [x] + ps])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not ops.RANGE_OUT.skip_next
