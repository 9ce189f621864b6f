"""ops.mha (batched MFMA GEMMs + masked softmax kernels through the C ABI) against torch.nn.MultiheadAttention
on the CPU in fp64 — the primitive mmcv's MultiheadAttention wraps: output, input gradients and
all parameter gradients, for the three mask layouts of the co-training step.  Tolerance 1e-3 (north star);
observed ~1e-6."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, ref):
    ref = ref.double()
    return float((a.detach().cpu().double() - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize('B,Lq,Lk,mask_kind', [(2, 100, 100, None), (2, 100, 256, 'image'), (2, 830, 830, 'shared'),
                                               (2, 37, 1024, 'head'), (1, 5, 64, 'image'), (2, 100, 4096, 'image'),
                                               (2, 800, 800, 'shared'), (1, 16, 2048, 'image')])
@pytest.mark.parametrize('core', [True, False])  # the fused core (csrc/attn_core.hip, the default) | q k^T -> softmax -> P v
def test_mha_matches_torch(cuda, B, Lq, Lk, mask_kind, core):
    from rscotr_amd import ops
    with ops.STATE.override(attn_core=core):
        _mha_against_torch(cuda, B, Lq, Lk, mask_kind)


def _mha_against_torch(cuda, B, Lq, Lk, mask_kind):
    from rscotr_amd import ops
    C, H = 256, 8
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    ref = torch.nn.MultiheadAttention(C, H, 0.0).double()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g).double() * 0.1)
    self_attn = Lq == Lk and mask_kind in (None, 'shared')
    q_in = torch.randn(B, Lq, C, generator=g)
    k_in = q_in if self_attn else torch.randn(B, Lk, C, generator=g)
    v_in = torch.randn(B, Lk, C, generator=g)
    ident = torch.randn(B, Lq, C, generator=g)
    mask = None
    if mask_kind == 'shared':
        mask = torch.rand(Lq, Lk, generator=g) < 0.4
        mask[:, 0] = False
        tmask = mask
    elif mask_kind == 'image':
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.5
        mask[:, :, 3] = False
        tmask = mask[:, None].expand(-1, H, -1, -1).reshape(B * H, Lq, Lk)
    elif mask_kind == 'head':
        mask = torch.rand(B * H, Lq, Lk, generator=g) < 0.5
        mask[:, :, 1] = False
        tmask = mask
    gy = torch.randn(B, Lq, C, generator=g)
    # reference (sequence-first, fp64)
    qr, kr, vr, ir = (t.double().clone().requires_grad_(True) for t in (q_in, k_in, v_in, ident))
    out_ref = ref(qr.transpose(0, 1), kr.transpose(0, 1), vr.transpose(0, 1), attn_mask=None if mask is None else tmask)[0]
    out_ref = out_ref.transpose(0, 1) + ir
    out_ref.backward(gy.double())
    # product
    qd = q_in.to(cuda).requires_grad_(True)
    kd = qd if self_attn else k_in.to(cuda).requires_grad_(True)
    vd, idd = v_in.to(cuda).requires_grad_(True), ident.to(cuda).requires_grad_(True)
    P = {n: p.detach().float().to(cuda).requires_grad_(True) for n, p in ref.named_parameters()}
    out = ops.mha(qd, kd, vd, P['in_proj_weight'], P['in_proj_bias'], P['out_proj.weight'], P['out_proj.bias'], H,
                  None if mask is None else mask.to(cuda), identity=idd)
    out.backward(gy.to(cuda))
    assert _rel(out, out_ref) < 1e-4
    if self_attn:
        assert _rel(qd.grad, qr.grad + kr.grad) < 1e-4
    else:
        assert _rel(qd.grad, qr.grad) < 1e-4 and _rel(kd.grad, kr.grad) < 1e-4
    assert _rel(vd.grad, vr.grad) < 1e-4 and _rel(idd.grad, ir.grad) < 1e-4
    for n, p in ref.named_parameters():
        assert _rel(P[n].grad, p.grad) < 1e-4, n

@pytest.mark.parametrize('B,H,Lq,Lk,mode', [(2, 8, 100, 4096, 2), (2, 8, 800, 800, 1), (1, 3, 45, 37, 3), (3, 2, 1, 1, 0),
                                            (1, 8, 33, 130, 2), (2, 4, 64, 1027, 0), (1, 8, 100, 16384, 2)])
def test_attn_core_against_fp64(cuda, B, H, Lq, Lk, mode):
    """rscotr_attn_core_fwd / _bwd through the C ABI against softmax(scale q k^T + mask) v in fp64 (strided q | k halves of one
    tensor, ragged lengths, every mask layout, key chunks, a fully blocked row -> zeros), twice: bit-identical results."""
    import ctypes
    from rscotr_amd._lib import lib
    hd, C = 32, H * 32
    g = torch.Generator().manual_seed(B * 1000 + Lq * 7 + Lk)
    self_attn = Lq == Lk
    ldq = 2 * C if self_attn else C
    qk = torch.randn(B, Lq, ldq, generator=g)
    kx = qk[..., C:] if self_attn else torch.randn(B, Lk, C, generator=g)
    q, k = qk[..., :C], kx
    v, do = torch.randn(B, Lk, C, generator=g), torch.randn(B, Lq, C, generator=g)
    mask = None
    if mode:
        shape = {1: (Lq, Lk), 2: (B, Lq, Lk), 3: (B * H, Lq, Lk)}[mode]
        mask = torch.rand(shape, generator=g) < 0.5
        mask[..., 0] = False
        if Lq > 2:
            mask[..., 2, :] = True  # a fully blocked row: output 0, no gradient through it
    scale = hd ** -0.5
    # fp64 reference
    q6, k6, v6 = (t.double().reshape(B, -1, H, hd).transpose(1, 2).clone().requires_grad_(True) for t in (q, k, v))
    s = (q6 @ k6.transpose(-1, -2)) * scale
    if mask is not None:
        m4 = mask[None, None] if mode == 1 else (mask[:, None] if mode == 2 else mask.reshape(B, H, Lq, Lk))
        s = s.masked_fill(m4, float('-inf'))
    p = torch.softmax(s, -1)
    p = torch.where(torch.isnan(p), torch.zeros_like(p), p)
    o6 = p @ v6
    o6.backward(do.double().reshape(B, Lq, H, hd).transpose(1, 2))
    unhead = lambda t, L: t.transpose(1, 2).reshape(B, L, C)
    dev = lambda t: t.contiguous().to(cuda)
    qk_d = dev(qk)
    k_d = None if self_attn else dev(k)
    v_d, do_d = dev(v), dev(do)
    m_d = None if mask is None else dev(mask)
    nws = lib.rscotr_attn_core_workspace(B, H, Lq, Lk)
    ws = torch.empty(max(nws, 16) // 4, dtype=torch.float32, device=cuda)
    st = torch.cuda.current_stream().cuda_stream
    kp = qk_d.data_ptr() + 4 * C if self_attn else k_d.data_ptr()
    runs = []
    for _ in range(2):
        out = torch.full((B, Lq, C), float('nan'), device=cuda)
        lse = torch.empty(B, H, Lq, device=cuda)
        dqk = torch.full((B, Lq, ldq), float('nan'), device=cuda)
        dk = None if self_attn else torch.full((B, Lk, C), float('nan'), device=cuda)
        dv = torch.full((B, Lk, C), float('nan'), device=cuda)
        mp = 0 if m_d is None else m_d.data_ptr()
        lib.call('rscotr_attn_core_fwd', qk_d.data_ptr(), kp, v_d.data_ptr(), mp, mode, out.data_ptr(), lse.data_ptr(), B, H, Lq,
                 Lk, hd, ldq, ldq, C, C, scale, ws.data_ptr(), nws, st)
        dkp = dqk.data_ptr() + 4 * C if self_attn else dk.data_ptr()
        lib.call('rscotr_attn_core_bwd', qk_d.data_ptr(), kp, v_d.data_ptr(), mp, mode, out.data_ptr(), do_d.data_ptr(),
                 lse.data_ptr(), dqk.data_ptr(), dkp, dv.data_ptr(), B, H, Lq, Lk, hd, ldq, ldq, C, C, ldq, ldq, C, scale,
                 ws.data_ptr(), nws, st)
        torch.cuda.synchronize()
        runs.append((out, lse, dqk, dv) + (() if self_attn else (dk,)))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    out, lse, dqk, dv = runs[0][:4]
    dq_p = dqk[..., :C]
    dk_p = dqk[..., C:] if self_attn else runs[0][4]
    def rel(a, ref):  # (a single key: dq = dk = 0 exactly in the reference, rounding noise here -> floor on the denominator)
        return float((a.cpu().double() - ref).abs().max() / max(float(ref.abs().max()), 1e-1))
    assert rel(out, unhead(o6.detach(), Lq)) < 2e-6
    assert rel(dq_p, unhead(q6.grad, Lq)) < 2e-5 and rel(dk_p, unhead(k6.grad, Lk)) < 2e-5 and rel(dv, unhead(v6.grad, Lk)) < 2e-5
    lse_ref = torch.logsumexp(s, -1)
    finite = torch.isfinite(lse_ref)
    assert torch.equal(torch.isfinite(lse.cpu()), finite)  # (+inf marks a fully blocked row)
    assert float((lse.cpu().double() - lse_ref)[finite].abs().max()) < 1e-5
    if mode and Lq > 2:
        assert float(out[:, 2].abs().max()) == 0.0


def test_attn_core_rejects_other_head_dims(cuda):
    from rscotr_amd._lib import lib
    t = torch.zeros(4096, device=cuda)
    with pytest.raises(RuntimeError, match='head dim'):
        lib.call('rscotr_attn_core_fwd', t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 0, t.data_ptr(), t.data_ptr(), 1, 1, 4, 4, 64,
                 64, 64, 64, 64, 0.125, 0, 0, torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('kind', ['self', 'self_const_pos', 'cross', 'cross_const_kpos', 'self_no_identity'])
def test_mha_positional_inputs_and_merged_gradients(cuda, kind):
    """The mmcv wrapper's positional adds and identity inside the node (ops.mha(x, kx, vx, ..., q_pos, k_pos, identity=x)):
    output and the gradients of x, q_pos, the key content and k_pos — merged in GEMM epilogues, no element-wise adds —
    against torch.nn.MultiheadAttention in fp64 with the adds written out."""
    from rscotr_amd import ops
    C, H, B, Lq, Lk = 256, 8, 2, 100, 320
    g = torch.Generator().manual_seed(11)
    ref = torch.nn.MultiheadAttention(C, H, 0.0).double()
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(torch.randn(p.shape, generator=g).double() * 0.1)
    self_attn = kind.startswith('self')
    x, qp = torch.randn(B, Lq, C, generator=g), torch.randn(B, Lq, C, generator=g)
    kx, kp = torch.randn(B, Lk, C, generator=g), torch.randn(B, Lk, C, generator=g)
    gy = torch.randn(B, Lq, C, generator=g)
    qp_grad = kind != 'self_const_pos'
    kp_grad = kind == 'cross'
    with_id = kind != 'self_no_identity'
    xr, qpr, kxr, kpr = (t.double().clone().requires_grad_(True) for t in (x, qp, kx, kp))
    qq = xr + qpr
    kk, vv = (qq, xr) if self_attn else (kxr + kpr, kxr)
    out_ref = ref(qq.transpose(0, 1), kk.transpose(0, 1), vv.transpose(0, 1))[0].transpose(0, 1)
    if with_id:
        out_ref = out_ref + xr
    out_ref.backward(gy.double())
    xd = x.to(cuda).requires_grad_(True)
    qpd = qp.to(cuda).requires_grad_(qp_grad)
    kxd = kx.to(cuda).requires_grad_(True)
    kpd = kp.to(cuda).requires_grad_(kp_grad)
    P = {n: p.detach().float().to(cuda).requires_grad_(True) for n, p in ref.named_parameters()}
    w = (P['in_proj_weight'], P['in_proj_bias'], P['out_proj.weight'], P['out_proj.bias'])
    if self_attn:
        out = ops.mha(xd, xd, xd, *w, H, None, identity=xd if with_id else None, q_pos=qpd, k_pos=qpd)
    else:
        out = ops.mha(xd, kxd, kxd, *w, H, None, identity=xd, q_pos=qpd, k_pos=kpd)
    out.backward(gy.to(cuda))
    assert _rel(out, out_ref) < 1e-4
    assert _rel(xd.grad, xr.grad) < 1e-4
    if qp_grad:
        assert _rel(qpd.grad, qpr.grad) < 1e-4
    if not self_attn:
        assert _rel(kxd.grad, kxr.grad) < 1e-4
        if kp_grad:
            assert _rel(kpd.grad, kpr.grad) < 1e-4
    for n, p in ref.named_parameters():
        assert _rel(P[n].grad, p.grad) < 1e-4, n


def test_mask_logits_matches_einsum(cuda):
    """einsum('bqd,bdhw->bqhw') of the seg head on the batched MFMA GEMM (token layout), with both gradients."""
    from rscotr_amd import ops
    g = torch.Generator().manual_seed(6)
    B, Q, D, P = 2, 100, 256, 4096
    e, mf, go = torch.randn(B, Q, D, generator=g), torch.randn(B, P, D, generator=g), torch.randn(B, Q, P, generator=g)
    er, mr = e.double().requires_grad_(True), mf.double().requires_grad_(True)
    ref = torch.einsum('bqd,bpd->bqp', er, mr)
    (ref * go.double()).sum().backward()
    ed, md = e.to(cuda).requires_grad_(True), mf.to(cuda).requires_grad_(True)
    out = ops.mask_logits(ed, md)
    (out * go.to(cuda)).sum().backward()
    assert _rel(out, ref) < 1e-5 and _rel(ed.grad, er.grad) < 1e-5 and _rel(md.grad, mr.grad) < 1e-5
