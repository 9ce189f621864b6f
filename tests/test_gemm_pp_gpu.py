"""The planes x planes split product (rscotr_split_planes / rscotr_gemm_pp, csrc/gemm_pp.hip) through the C ABI: the plane
sets bit for bit against a NumPy / torch restatement of the ST32 layout (include/rscotr.h), the product in its four operand
modes against fp64 references of the same contractions (tolerance 1e-5 of max|C|: the six-term product is fp32-FMA class,
measured 3e-7; the north star's bound is 1e-3), epilogues, k-slices, the deferred combine, and the routing inside ops.gemm."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.double()
    return float((a.detach().cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))


def st32_reference(x):
    """(rows, cols) fp32 CPU tensor -> uint16 array of the plane set: x = h + m + l with h = bf16(x), m = bf16(x - h),
    l = bf16(x - h - m) (round to nearest even, both subtractions exact in fp32); 32 x 32 super-tiles in row-major order, three
    planes of 1024 values per tile, value (r, c) of a tile at 8 * slot(r, c // 8) + c % 8,
    slot(r, c8) = 32 c8 + 16 (r // 16) + 4 ((r // 4 + c8) % 4) + r % 4."""
    R, C = x.shape
    RT, CT = (R + 31) // 32, (C + 31) // 32
    xp = torch.zeros(RT * 32, CT * 32)
    xp[:R, :C] = x
    h = xp.to(torch.bfloat16)
    r1 = xp - h.float()
    m = r1.to(torch.bfloat16)
    l = (r1 - m.float()).to(torch.bfloat16)
    planes = torch.stack([h, m, l]).view(torch.int16).numpy().astype(np.uint16)       # (3, RT*32, CT*32)
    out = np.zeros((RT, CT, 3, 1024), dtype=np.uint16)
    r = np.arange(32)[:, None]
    c = np.arange(32)[None, :]
    c8 = c // 8
    pos = 8 * (32 * c8 + 16 * (r // 16) + 4 * ((r // 4 + c8) % 4) + r % 4) + c % 8   # (32, 32)
    t = planes.reshape(3, RT, 32, CT, 32).transpose(1, 3, 0, 2, 4)                     # (RT, CT, 3, 32, 32)
    out[..., pos.reshape(-1)] = t.reshape(RT, CT, 3, 1024)
    return out.reshape(-1)


@pytest.mark.parametrize('R,C,ld', [(32, 32, 32), (37, 45, 45), (300, 200, 208), (1000, 96, 96), (257, 264, 264), (10880, 256, 256),
                                    (70, 33, 36)])
def test_split_planes_layout_bit_exact(cuda, R, C, ld):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, ld, generator=g) * torch.exp(torch.randn(R, ld, generator=g) * 3)   # a wide range of exponents
    ps, _ = ops.split_planes(x.to(cuda), R, C, ld)
    got = ps.buf.cpu().view(torch.int16).numpy().astype(np.uint16)
    want = st32_reference(x[:, :C].contiguous())
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    # the three planes carry all 24 significand bits: their fp32 sum IS the input
    pl = torch.from_numpy(got.astype(np.int16)).view(torch.bfloat16).float().reshape(-1, 3, 1024)
    back = (pl[:, 0] + pl[:, 1]) + pl[:, 2]
    ref = torch.from_numpy(st32_reference(x[:, :C].contiguous()).astype(np.int16)).view(torch.bfloat16).float().reshape(-1, 3, 1024)
    assert torch.equal(back, (ref[:, 0] + ref[:, 1]) + ref[:, 2])


def test_split_planes_rowscale_and_column_sums(cuda):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(5)
    R, C, per = 1200, 136, 300
    x = torch.randn(R, C, generator=g)
    sc = torch.tensor([0.0, 1.25, 1.0, 2.0 / 3.0])
    ps, parts = ops.split_planes(x.to(cuda), R, C, C, rowscale=sc.to(cuda), rows_per=per, colsum=True)
    xs = x * sc.repeat_interleave(per)[:, None]
    assert np.array_equal(ps.buf.cpu().view(torch.int16).numpy().astype(np.uint16), st32_reference(xs))
    assert parts.shape == ((R + 255) // 256, C)
    assert _rel(parts.sum(0), xs.double().sum(0)) < 1e-6
    for gidx in range(parts.shape[0]):
        assert _rel(parts[gidx], xs[gidx * 256:(gidx + 1) * 256].double().sum(0)) < 1e-6


class _route:
    """The planes x planes route inside ops.gemm for the duration of a test (it is opt-in: RSCOTR_PP=1), with its size rules
    opened up so that test-sized problems take it; everything is put back on exit (tests/conftest.py checks)."""

    def __enter__(self):
        from rscotr_amd import ops
        self.keep = {k: getattr(ops.PP, k) for k in ('enabled', 'min_tiles', 'min_n', 'max_split', 'min_rows', 'min_work')}
        ops.PP.enabled, ops.PP.min_tiles, ops.PP.min_n, ops.PP.max_split = True, 1, 64, 1 << 30
        ops.PP.clear()
        return ops.PP

    def __exit__(self, *exc):
        from rscotr_amd import ops
        for k, v in self.keep.items():
            setattr(ops.PP, k, v)
        ops.PP.clear()
        return False


def _operands(M, N, K, ac, bc, seed, cuda):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((K, M) if ac else (M, K), generator=g)
    B = torch.randn((K, N) if bc else (N, K), generator=g)
    ref = (A.double().t() if ac else A.double()) @ (B.double() if bc else B.double().t())
    return A, B, ref


@pytest.mark.parametrize('M,N,K', [(128, 128, 16), (128, 128, 96), (130, 96, 96), (257, 288, 100), (64, 64, 256), (500, 128, 2048),
                                   (1000, 768, 352), (33, 31, 19), (2048, 384, 384), (10880, 256, 256), (256, 2048, 1600)])
@pytest.mark.parametrize('ac,bc', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_pp_modes(cuda, M, N, K, ac, bc):
    from rscotr_amd import ops
    A, B, ref = _operands(M, N, K, ac, bc, M * 7 + N * 3 + K + ac * 2 + bc, cuda)
    pa, _ = ops.split_planes(A.to(cuda), A.shape[0], A.shape[1], A.shape[1])
    pb, _ = ops.split_planes(B.to(cuda), B.shape[0], B.shape[1], B.shape[1])
    out = torch.full((M, N), float('nan'), device=cuda)
    ops.gemm_pp(pa, ac, pb, bc, M, N, K, out=out)
    assert _rel(out, ref) < 1e-5


def test_gemm_pp_is_fp32_accurate(cuda):
    """The error class of the product: against fp64, no worse than 1.5 x an fp32 FMA chain (torch.mm in fp32 on the CPU)."""
    from rscotr_amd import ops
    M, N, K = 1024, 512, 2048
    A, B, ref = _operands(M, N, K, 0, 0, 11, cuda)
    pa, _ = ops.split_planes(A.to(cuda), M, K, K)
    pb, _ = ops.split_planes(B.to(cuda), N, K, K)
    e_pp = _rel(ops.gemm_pp(pa, 0, pb, 0, M, N, K), ref)
    e_f32 = _rel(A @ B.t(), ref)
    assert e_pp <= max(1e-6, 1.5 * e_f32), (e_pp, e_f32)


@pytest.mark.parametrize('act', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('M,N,K', [(300, 200, 128), (384, 256, 1024), (130, 77, 256)])
def test_gemm_pp_epilogues(cuda, act, M, N, K):
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(act)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, aux, resid, c0 = (torch.randn(N, generator=g), torch.randn(M, N, generator=g),
                            torch.randn(M, N, generator=g), torch.randn(M, N, generator=g))
    rs = torch.rand(3, generator=g) + 0.5
    per = (M + 2) // 3
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
    v = v * rs.double().repeat_interleave(per)[:M, None]
    ref = v + resid.double() + c0.double()
    pa, _ = ops.split_planes(A.to(cuda), M, K, K)
    pb, _ = ops.split_planes(B.to(cuda), N, K, K)
    out = c0.clone().to(cuda)
    pre = torch.empty(M, N, device=cuda)
    ops.gemm_pp(pa, 0, pb, 0, M, N, K, out=out, bias=bias.to(cuda), act=act, aux=aux.to(cuda), pre=pre, resid=resid.to(cuda),
                accumulate=True, rowscale=rs.to(cuda), rows_per=per)
    assert _rel(out, ref) < 1e-5
    assert _rel(pre, pre_ref) < 1e-5
    # second output: C without the residual, C2 = C + resid
    out = c0.clone().to(cuda)
    out2 = torch.full((M, N), float('nan'), device=cuda)
    ops.gemm_pp(pa, 0, pb, 0, M, N, K, out=out, resid=resid.to(cuda), accumulate=True, out2=out2)
    base = A.double() @ B.double().t() + c0.double()
    assert _rel(out, base) < 1e-5 and _rel(out2, base + resid.double()) < 1e-5


@pytest.mark.parametrize('M,N,K', [(256, 2048, 10880), (2048, 256, 10880), (384, 1536, 2048), (288, 96, 32768)])
def test_weight_gradient_route_with_bias_gradient_and_sample_scale(cuda, M, N, K):
    """dW = (s dy)^T x with db = column sums of s dy through ops.gemm (a_kmajor = b_kmajor = 1, rowsum, kscale): the planes x
    planes route splits s * dy once, takes the bias gradient from that pass, and cuts the reduction into k-slices."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(M + N)
    dy, x = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    nb = 4
    s = torch.tensor([1.0, 0.0, 1.25, 1.0 / 0.9])
    per = K // nb
    dys = dy.double() * s.double().repeat_interleave(per)[:, None]
    ref_w, ref_b = dys.t() @ x.double(), dys.sum(0)
    before = dict(ops.PP.stats)
    rs = torch.zeros(M, device=cuda)
    with _route():
        out = ops.gemm(dy.to(cuda), x.to(cuda), M, N, K, M, N, 1, 1, rowsum=rs, kscale=s.to(cuda), krows_per=per)
    assert ops.PP.stats['products'] == before['products'] + 1, 'the product did not take the planes x planes route'
    assert _rel(out, ref_w) < 1e-5 and _rel(rs, ref_b) < 1e-5


def test_route_shares_plane_sets_between_the_products_of_a_linear(cuda):
    """y = x W^T, dx = dy W, dW = dy^T x through ops.gemm: x and dy are split ONCE each (W twice here: without an optimizer arena
    a parameter is an ordinary tensor), and every product equals its fp64 reference."""
    from rscotr_amd import ops
    M, N, K = 4096, 512, 256
    g = torch.Generator().manual_seed(3)
    x, W, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(M, N, generator=g)
    xd, Wd, dyd = x.to(cuda), W.to(cuda), dy.to(cuda)
    with _route():
        s0 = dict(ops.PP.stats)
        y = ops.gemm(xd, Wd, M, N, K, K, K, 0, 0)
        dx = ops.gemm(dyd, Wd, M, K, N, N, K, 0, 1)
        dW = ops.gemm(dyd, xd, N, K, M, N, K, 1, 1)
        s1 = ops.PP.stats
        assert s1['products'] - s0['products'] == 3 and s1['splits'] - s0['splits'] == 3 and s1['hits'] - s0['hits'] == 3
        assert _rel(y, x.double() @ W.double().t()) < 1e-5
        assert _rel(dx, dy.double() @ W.double()) < 1e-5
        assert _rel(dW, dy.double().t() @ x.double()) < 1e-5
        # an in-place change of an operand is seen (tensor version): no stale planes
        xd.mul_(2.0)
        y2 = ops.gemm(xd, Wd, M, N, K, K, K, 0, 0)
        assert _rel(y2, 2.0 * x.double() @ W.double().t()) < 1e-5
