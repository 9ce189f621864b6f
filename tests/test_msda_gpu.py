"""MSDA HIP kernels (through the C ABI) vs the oracle on the same seeded inputs."""
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['tiled', 'sorted', 'scatter'], autouse=True)
def bwd_strategy(request):
    """Both grad_value strategies of rscotr_msda_bwd run every test (include/rscotr.h)."""
    from rscotr_amd import ops
    with ops.STATE.override(msda_bwd=request.param):
        yield request.param

SHAPES_512 = [(64, 64), (32, 32), (16, 16), (8, 8)]


def _inputs(B, shapes, Nq, H, D, P, seed, spread=0.15, dev='cpu'):
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    Nk = sum(h * w for h, w in shapes)
    value = torch.randn(B, Nk, H, D, generator=g)
    ref = torch.rand(B, Nq, 1, 1, 1, 2, generator=g)
    loc = ref + spread * torch.randn(B, Nq, H, L, P, 2, generator=g)  # some land outside [0,1]
    attn = torch.softmax(torch.randn(B, Nq, H, L * P, generator=g), -1).view(B, Nq, H, L, P)
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
    return value, ss, lsi, loc, attn


def _run_pair(value, ss, lsi, loc, attn, cuda):
    from rscotr_amd import ops
    # oracle (CPU)
    v0, l0, a0 = (t.clone().requires_grad_(True) for t in (value, loc, attn))
    out0 = O.msda_sample(v0, ss, lsi, l0, a0)
    go = torch.randn(out0.shape, generator=torch.Generator().manual_seed(7))
    out0.backward(go)
    # HIP
    v1, l1, a1 = (t.clone().to(cuda).requires_grad_(True) for t in (value, loc, attn))
    out1 = ops.msda(v1, ss.to(cuda), lsi.to(cuda), l1, a1)
    out1.backward(go.to(cuda))
    torch.cuda.synchronize()
    return (out0, v0.grad, l0.grad, a0.grad), (out1.cpu(), v1.grad.cpu(), l1.grad.cpu(), a1.grad.cpu())


def _close(a, b, rtol=1e-3, atol=None):
    # tolerance from BASELINE.json north_star: 1e-3 relative, fp32
    atol = atol if atol is not None else 1e-3 * float(a.abs().max()) + 1e-6
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


@pytest.mark.parametrize('B,Nq,H,D,P', [(2, 100, 8, 32, 4), (1, 37, 8, 32, 4), (2, 65, 4, 16, 2),
                                        (1, 50, 2, 64, 8), (3, 33, 8, 32, 1)])
def test_msda_small(cuda, B, Nq, H, D, P):
    shapes = [(12, 9), (6, 5), (3, 3), (2, 1)]
    ref, got = _run_pair(*_inputs(B, shapes, Nq, H, D, P, seed=B * 100 + Nq), cuda)
    for r, g in zip(ref, got):
        _close(r, g)


def test_msda_encoder_shape_512(cuda):
    """configs[1] encoder call: B=2, Nq=Nk=5440, 8 heads x 32, 4 levels x 4 points."""
    N = sum(h * w for h, w in SHAPES_512)
    ref, got = _run_pair(*_inputs(2, SHAPES_512, N, 8, 32, 4, seed=1), cuda)
    for r, g in zip(ref, got):
        _close(r, g)


def test_msda_collisions_and_borders(cuda):
    """Every query samples the same few spots (long per-token lists -> multi-chunk combine in the
    sorted strategy), plus samples on the map border / one pixel outside (extended-grid bins)."""
    shapes = [(8, 8), (4, 4), (2, 2), (1, 1)]
    value, ss, lsi, loc, attn = _inputs(2, shapes, 300, 8, 32, 4, seed=5)
    loc = loc.clone()
    loc[:, :, :, :, 0] = 0.5                       # all queries hit the map centre
    loc[:, :, :, :, 1, 0] = 0.0                    # x on the left border (top-left tap outside)
    loc[:, :, :, :, 2] = 1.0                       # bottom-right border
    loc[:, ::7, :, :, 3] = -0.3                    # fully outside
    ref, got = _run_pair(value, ss, lsi, loc, attn, cuda)
    for r, g in zip(ref, got):
        _close(r, g)


def test_msda_decoder_shape(cuda):
    """DINO decoder cross-attention shape: Nq = 800 queries against Nk = 5440 tokens."""
    ref, got = _run_pair(*_inputs(2, SHAPES_512, 800, 8, 32, 4, seed=2), cuda)
    for r, g in zip(ref, got):
        _close(r, g)


def test_msda_known_answers(cuda):
    from rscotr_amd import ops
    shapes = [(8, 8), (4, 4)]
    ss = torch.tensor(shapes, dtype=torch.long)
    lsi = torch.tensor([0, 64])
    B, H, D, L, P = 1, 8, 32, 2, 4
    Nk = 80
    # constant map -> constant * sum(weights) for interior samples
    value = torch.full((B, Nk, H, D), 3.0)
    loc = torch.full((B, 5, H, L, P, 2), 0.5)
    attn = torch.full((B, 5, H, L, P), 1.0 / (L * P))
    out = ops.msda(value.to(cuda), ss.to(cuda), lsi.to(cuda), loc.to(cuda), attn.to(cuda)).cpu()
    assert torch.allclose(out, torch.full_like(out, 3.0), atol=1e-6)
    # sampling exactly at a pixel centre returns the pixel
    value = torch.randn(B, Nk, H, D)
    loc = torch.zeros(B, 1, H, L, P, 2)
    loc[..., 0] = (3 + 0.5) / 8  # x = col 3 on level 0
    loc[..., 1] = (5 + 0.5) / 8  # y = row 5
    attn = torch.zeros(B, 1, H, L, P)
    attn[:, :, :, 0, 0] = 1.0
    out = ops.msda(value.to(cuda), ss.to(cuda), lsi.to(cuda), loc.to(cuda), attn.to(cuda)).cpu()
    assert torch.allclose(out.view(H, D), value[0, 5 * 8 + 3], atol=1e-6)
    # everything outside the map -> zeros, and NaN-free
    loc = torch.full((B, 3, H, L, P, 2), 7.0)
    attn = torch.full((B, 3, H, L, P), 0.125)
    out = ops.msda(value.to(cuda), ss.to(cuda), lsi.to(cuda), loc.to(cuda), attn.to(cuda)).cpu()
    assert torch.equal(out, torch.zeros_like(out))


def test_msda_empty_and_errors(cuda):
    from rscotr_amd import ops
    ss = torch.tensor([(4, 4)], dtype=torch.long, device=cuda)
    lsi = torch.zeros(1, dtype=torch.long, device=cuda)
    value = torch.randn(1, 16, 8, 32, device=cuda)
    out = ops.msda(value, ss, lsi, torch.zeros(1, 0, 8, 1, 4, 2, device=cuda), torch.zeros(1, 0, 8, 1, 4, device=cuda))
    assert out.shape == (1, 0, 256)
    with pytest.raises(RuntimeError):  # unsupported head width -> RSCOTR_E_SHAPE, raised loudly
        ops.msda(torch.randn(1, 16, 8, 24, device=cuda), ss, lsi,
                 torch.zeros(1, 2, 8, 1, 4, 2, device=cuda), torch.zeros(1, 2, 8, 1, 4, device=cuda))
    with pytest.raises(RuntimeError):  # CPU tensors are rejected: no fallback
        ops.msda(value.cpu(), ss.cpu(), lsi.cpu(), torch.zeros(1, 2, 8, 1, 4, 2), torch.zeros(1, 2, 8, 1, 4))


def test_msda_linearity_full_size(cuda):
    """Size-independent property at the 800x800 det shape (configs[3], N=13294, B=4):
    out(a*v1 + v2) == a*out(v1) + out(v2) for fixed locations/weights."""
    from rscotr_amd import ops
    shapes = [(100, 100), (50, 50), (25, 25), (13, 13)]
    value, ss, lsi, loc, attn = _inputs(4, shapes, 13294, 8, 32, 4, seed=3)
    g = torch.Generator().manual_seed(11)
    v2 = torch.randn(value.shape, generator=g)
    dv = [t.to(cuda) for t in (value, v2, loc, attn)]
    ssd, lsid = ss.to(cuda), lsi.to(cuda)
    o1 = ops.msda(dv[0], ssd, lsid, dv[2], dv[3])
    o2 = ops.msda(dv[1], ssd, lsid, dv[2], dv[3])
    o12 = ops.msda(2.5 * dv[0] + dv[1], ssd, lsid, dv[2], dv[3])
    assert torch.allclose(o12, 2.5 * o1 + o2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('refdim,L,P', [(2, 4, 4), (4, 4, 4), (2, 3, 2), (4, 1, 8)])
def test_msda_prep_matches_torch(cuda, refdim, L, P):
    """Fused prologue (softmax over L*P, loc from reference points and offsets) against the formulas of mmcv
    MultiScaleDeformableAttention.forward in fp64, with gradients of the offsets and the logits."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(refdim * 10 + L + P)
    B, Nq, H = 2, 37, 8
    off = torch.randn(B, Nq, H * L * P * 2, generator=g)
    logit = torch.randn(B, Nq, H, L * P, generator=g)
    ref = torch.rand(B, Nq, L, refdim, generator=g) * 0.8 + 0.1
    norm = torch.tensor([[7.0 * (l + 1), 5.0 * (l + 1)] for l in range(L)])
    gl = torch.randn(B, Nq, H, L, P, 2, generator=g)
    ga = torch.randn(B, Nq, H, L, P, generator=g)
    o, lg, r, nm = off.double().requires_grad_(True), logit.double().requires_grad_(True), ref.double(), norm.double()
    o6 = o.view(B, Nq, H, L, P, 2)
    if refdim == 2:
        loc_r = r[:, :, None, :, None, :] + o6 / nm[None, None, None, :, None, :]
    else:
        loc_r = r[:, :, None, :, None, :2] + o6 / P * r[:, :, None, :, None, 2:] * 0.5
    aw_r = lg.softmax(-1).view(B, Nq, H, L, P)
    ((loc_r * gl.double()).sum() + (aw_r * ga.double()).sum()).backward()
    od, ld = off.to(cuda).requires_grad_(True), logit.to(cuda).requires_grad_(True)
    loc, aw = ops.msda_prep(od, ld, ref.to(cuda), norm.to(cuda), L, P)
    ((loc * gl.to(cuda)).sum() + (aw * ga.to(cuda)).sum()).backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(loc, loc_r) < 1e-6 and rel(aw, aw_r) < 1e-6
    assert rel(od.grad, o.grad) < 1e-5 and rel(ld.grad, lg.grad) < 1e-5


def test_sine_embed4_matches_reference_formula(cuda):
    """Fused decoder position embedding vs the step-by-step formula of transformer.py:43-76 (kept as
    DinoTransformerDecoder.gen_sineembed_for_position)."""
    from rscotr_amd import ops
    from rscotr_amd.det_head import DinoTransformerDecoder
    pos = torch.rand(2, 77, 4, generator=torch.Generator().manual_seed(4))
    ref = DinoTransformerDecoder.gen_sineembed_for_position(pos.double())
    out = ops.sine_embed4(pos.to(cuda)).cpu().double()
    assert out.shape == ref.shape and float((out - ref).abs().max()) < 2e-4  # fp32 sin/cos of arguments up to 2*pi


@pytest.mark.parametrize('shapes,Nq', [
    ([(100, 100), (50, 50), (25, 25), (13, 13)], 13294),     # BASELINE configs[3] (det 800^2): the encoder call, Nq = Nk
    ([(128, 128), (64, 64), (32, 32), (16, 16)], 21760),     # BASELINE configs[4] (Swin-B 1024^2): Nq = Nk = 21760
    ([(1, 30000), (2, 2)], 4000),                             # degenerate pyramid: 60 011 bins exceed the LDS histogram
])
def test_msda_large_pyramids(cuda, bwd_strategy, shapes, Nq):
    """The pyramids of BASELINE configs[3] / configs[4] at their full token counts, and one whose host-side bin bound
    (2 Nk + 2 L + 2) exceeds the LDS histogram of the sorted strategy (the kernels decide on the device from the level
    shapes whether that path runs or stands down for the atomic scatter: csrc/msda.hip, MSDA_LDS_WORDS): every strategy
    against the ORACLE (seg_head/pixel_decoder.py:134-146, bbox_head/transformer.py:211-221 reach the op at these shapes)."""
    L = len(shapes)
    value, ss, lsi, loc, attn = _inputs(1, shapes, Nq, 8, 32, 4, seed=11, spread=0.05)
    ref, got = _run_pair(value, ss, lsi, loc, attn, cuda)
    for r, g in zip(ref, got):
        assert torch.isfinite(g).all()
        _close(r, g)


@pytest.mark.parametrize('kind', ['encoder', 'decoder', 'decoder_const_pos', 'plain'])
def test_msda_attention_block_matches_composition(cuda, kind):
    """ops.msda_attention (the whole mmcv MultiScaleDeformableAttention.forward as one node, gradients merged in GEMM
    epilogues) against the composition of the single ops (linear / msda_prep / msda with autograd's own adds): output,
    input and parameter gradients; then the composition itself is what tests/test_model_gpu.py holds to the oracle."""
    from rscotr_amd import ops
    torch.manual_seed(3)
    B, H, L, P, C = 2, 8, 4, 4, 256
    shapes = [(16, 16), (8, 8), (4, 4), (2, 2)]
    Nk = sum(h * w for h, w in shapes)
    Nq = Nk if kind == 'encoder' else 90
    ss = torch.tensor(shapes, dtype=torch.int64, device=cuda)
    lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    norm = torch.stack([ss[:, 1], ss[:, 0]], -1).float()
    refdim = 2 if kind == 'encoder' else 4
    ref = torch.rand(B, Nq, L, refdim, device=cuda) * 0.8 + 0.1
    mk = lambda *s, sc=1.0: (torch.randn(*s, device=cuda) * sc)
    W = dict(w_off=mk(H * L * P * 2, C, sc=0.02), b_off=mk(H * L * P * 2), w_aw=mk(H * L * P, C, sc=0.05), b_aw=mk(H * L * P, sc=0.1),
             w_v=mk(C, C, sc=0.06), b_v=mk(C, sc=0.1), w_o=mk(C, C, sc=0.06), b_o=mk(C, sc=0.1))
    x0, pos0, mem0, gy = mk(B, Nq, C), mk(B, Nq, C), mk(B, Nk, C), mk(B, Nq, C)
    pos_grad = kind in ('encoder', 'decoder')
    res = []
    for fused in (False, True):
        x, pos, mem = x0.clone().requires_grad_(True), pos0.clone().requires_grad_(pos_grad), mem0.clone().requires_grad_(True)
        Wp = {k: v.clone().requires_grad_(True) for k, v in W.items()}
        value = x if kind == 'encoder' else mem
        ident = None if kind == 'plain' else x
        if fused:
            y = ops.msda_attention(x, pos, value, ident, None, ref, ss, lsi, norm, H, L, P, Wp['w_off'], Wp['b_off'],
                                   Wp['w_aw'], Wp['b_aw'], Wp['w_v'], Wp['b_v'], Wp['w_o'], Wp['b_o'])
        else:
            q = x + pos
            v = ops.linear(value, Wp['w_v'], Wp['b_v']).view(B, Nk, H, C // H)
            off = ops.linear(q, Wp['w_off'], Wp['b_off'])
            aw = ops.linear(q, Wp['w_aw'], Wp['b_aw']).view(B, Nq, H, L * P)
            loc, a = ops.msda_prep(off, aw, ref, norm, L, P)
            y = ops.linear(ops.msda(v, ss, lsi, loc, a), Wp['w_o'], Wp['b_o'], resid=ident)
        y.backward(gy)
        res.append((y, x.grad, pos.grad if pos_grad else None, None if kind == 'encoder' else mem.grad,
                    *[Wp[k].grad for k in sorted(Wp)]))
    for a, b in zip(*res):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-9, (kind, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize('refdim,packed', [(2, True), (2, False), (4, True)])
@pytest.mark.parametrize('Nq', [5440, 1100, 37])
def test_fused_prologue_equals_the_kernel_pair(cuda, refdim, packed, Nq):
    """rscotr_msda_fwd_prep (the softmax / location prologue by the threads that stage the samples) against
    rscotr_msda_prep_fwd + rscotr_msda_fwd: loc, attn and the output bit for bit (mmcv MultiScaleDeformableAttention.forward,
    bbox_head/transformer.py:211-221,258-269)."""
    from rscotr_amd import ops
    from rscotr_amd.ops import deform
    shapes = [(64, 64), (32, 32), (16, 16), (8, 8)]
    B, H, D, L, P = 2, 8, 32, 4, 4
    Nk = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(Nq + refdim)
    ss = torch.tensor(shapes, dtype=torch.long, device=cuda)
    lsi = torch.tensor([0, 4096, 5120, 5376], dtype=torch.long, device=cuda)
    value = torch.randn((B, Nk, H, D), generator=g).to(cuda)
    n = H * L * P
    both = (torch.randn((B * Nq, 3 * n), generator=g) * 2.0).to(cuda)
    if packed:
        off, logit, ldo, ldl = both, both.view(-1)[2 * n:], 3 * n, 3 * n
    else:
        off, logit, ldo, ldl = both[:, :2 * n].contiguous(), both[:, 2 * n:].contiguous(), 2 * n, n
    ref = torch.rand((B, Nq, L if refdim == 2 else 1, refdim), generator=g).to(cuda)
    norm = torch.tensor([(w, h) for h, w in shapes], dtype=torch.float32, device=cuda) if refdim == 2 else None
    assert deform._msda_fused_ok(Nk, H, D, L, P)
    loc0, attn0 = deform._msda_prep_fwd_raw(off, logit, ref, norm, B, Nq, H, L, P, ld_off=ldo, ld_logit=ldl)
    out0 = deform._msda_fwd_raw(value, ss, lsi, loc0, attn0)
    loc1, attn1, out1 = deform._msda_fwd_prep_raw(value, ss, lsi, off, logit, ref, norm, L, P, ldo, ldl)
    assert torch.equal(loc0, loc1) and torch.equal(attn0, attn1) and torch.equal(out0, out1)
    assert float(out1.abs().max()) > 0
