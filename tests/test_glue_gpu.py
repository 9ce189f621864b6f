"""Glue kernels between the attention blocks (csrc/glue.hip) against plain PyTorch on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,sizes,C,rows,row0,batched', [(2, (64, 256, 1024, 4096), 256, 4, 0, False), (2, (4096,), 256, 3, 2, False),
                                                         (3, (5, 1, 7), 8, 5, 1, True), (1, (33,), 12, 1, 0, True)])
def test_level_embed_add_and_gradient(cuda, B, sizes, C, rows, row0, batched):
    """ops.level_embed_add == x + const + weight[row0 + level] (transformer.py:196-207, pixel_decoder.py:108-118,
    mask2former_head.py:152-156); gradient of the rows = per-level sums (fp64 reference), bit-identical between two runs;
    x read through a batch-strided view."""
    from rscotr_amd import ops
    g = torch.Generator(device='cpu').manual_seed(3)
    N = sum(sizes)
    big = torch.randn(B, N + 5, C, generator=g).to(cuda)
    w = torch.randn(rows, C, generator=g).to(cuda).requires_grad_(True)
    cst = torch.randn(B if batched else 1, N, C, generator=g).to(cuda)
    lv = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(cuda) + row0
    for use_x in (True, False):
        w.grad = None
        xin = big[:, 2:2 + N] if use_x else None  # batch stride (N + 5) * C, dense rows
        out = ops.level_embed_add(xin, w, sizes, const=cst, batch=B, row0=row0)
        ref = cst + w.detach()[lv][None]
        if use_x:
            ref = xin + ref
        assert out.shape == (B, N, C) and float((out - ref.expand(B, N, C)).abs().max()) <= 1e-6 * float(ref.abs().max())
        go = torch.randn(B, N, C, generator=g).to(cuda)
        out.backward(go)
        dref = torch.zeros(rows, C, dtype=torch.float64, device=cuda).index_add_(0, lv.repeat(B), go.double().reshape(-1, C))
        assert float((w.grad.double() - dref).abs().max()) <= 1e-5 * max(1.0, float(dref.abs().max()))
        first = w.grad.clone()
        w.grad = None
        ops.level_embed_add(xin, w, sizes, const=cst, batch=B, row0=row0).backward(go)
        assert torch.equal(first, w.grad)


@pytest.mark.parametrize('refdim', [2, 4])
def test_msda_prep_strided_rows_and_shared_reference_points(cuda, refdim):
    """rscotr_msda_prep_fwd / _bwd with the offsets | logits read from (written to) the column blocks of ONE (B*Nq, 3n)
    tensor and with reference points shared by the levels (ref_levels = 1) give bit-identical results to the dense,
    per-level form (mmcv MultiScaleDeformableAttention.forward's element-wise lines)."""
    from rscotr_amd import ops
    B, Nq, H, L, P = 2, 37, 8, 4, 4
    n = H * L * P
    g = torch.Generator(device='cpu').manual_seed(5)
    both = torch.randn(B * Nq, 3 * n, generator=g).to(cuda)
    off, logit = both[:, :2 * n].contiguous(), both[:, 2 * n:].contiguous()
    ref1 = torch.rand(B, Nq, 1, refdim, generator=g).to(cuda)
    refL = ref1.expand(B, Nq, L, refdim).contiguous()
    norm = torch.tensor([[64., 64.], [32., 32.], [16., 16.], [8., 8.]], device=cuda)
    loc0, attn0 = ops._msda_prep_fwd_raw(off, logit, refL, norm, B, Nq, H, L, P)
    loc1, attn1 = ops._msda_prep_fwd_raw(both, both.view(-1)[2 * n:], ref1, norm, B, Nq, H, L, P, ld_off=3 * n, ld_logit=3 * n)
    assert torch.equal(loc0, loc1) and torch.equal(attn0, attn1)
    gloc, gattn = torch.randn(loc0.shape, generator=g).to(cuda), torch.randn(attn0.shape, generator=g).to(cuda)
    goff0, glogit0 = ops._msda_prep_bwd_raw(gloc, gattn, attn0, refL, norm, B, Nq, H, L, P)
    packed, none = ops._msda_prep_bwd_raw(gloc, gattn, attn0, ref1, norm, B, Nq, H, L, P, packed=True)
    assert none is None and packed.shape == (B * Nq, 3 * n)
    assert torch.equal(packed[:, :2 * n], goff0.view(B * Nq, 2 * n)) and torch.equal(packed[:, 2 * n:], glogit0.view(B * Nq, n))


def test_pack4(cuda):
    from rscotr_amd import ops
    from rscotr_amd._lib import lib
    parts = [torch.randn(k, device=cuda) for k in (1000, 7, 0, 333)]
    out = torch.empty(1340, device=cuda)
    slot = ops.RANGES.new_slot(cuda)
    lib.call('rscotr_pack4', parts[0].data_ptr(), 1000, parts[1].data_ptr(), 7, 0, 0, parts[3].data_ptr(), 333, out.data_ptr(),
             slot, ops._stream())
    assert torch.equal(out, torch.cat(parts))
    # the range word of the packed tensor rides along (include/rscotr.h: rscotr_gemm_f32_r)
    lo, hi = ops.RANGES.word(slot)
    assert lo <= float(out.abs().max()) < hi


@pytest.mark.parametrize('uniform', [True, False])
def test_cdn_queries_kernel(cuda, uniform):
    """ops.cdn_queries (one launch) against the op-by-op PyTorch form of the same slot arithmetic
    (query_denoising.py:104-178) on the same random numbers; embedding gradient against autograd's, twice (bitwise)."""
    from rscotr_amd import ops
    from util import cdn_queries_ref
    B, PC, G, C, NC = 2, 200, 32, 256, 20
    g = torch.Generator(device='cpu').manual_seed(11)
    w = torch.randn(NC, C, generator=g).to(cuda).requires_grad_(True)
    gt_lab = torch.randint(0, NC, (B * G,), generator=g).to(cuda)
    cxcy = torch.rand(B * G, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(B * G, 2, generator=g) * 0.3 + 0.01
    gt_boxn = torch.cat([cxcy, wh], -1).to(cuda)
    slot_src = torch.randint(0, B * G, (B, PC), generator=g).to(cuda)
    slot_valid = (torch.rand(B, PC, generator=g) < 0.7).float().to(cuda)
    slot_neg = (torch.arange(PC) // 8 % 2).float()[None].expand(B, PC).contiguous().to(cuda)
    u = torch.rand(B, PC, 10, generator=g)
    if not uniform:
        u[..., 1] = torch.randint(0, NC, (B, PC), generator=g).float()
        u[..., 2:6] = torch.randint(0, 2, (B, PC, 4), generator=g).float()
    u = u.to(cuda)
    ql, qb = ops.cdn_queries(w, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, 0.5, 0.4, NC)
    wr = w.detach().clone().requires_grad_(True)
    rl, rb = cdn_queries_ref(wr, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, 0.5, 0.4, NC)
    assert torch.equal(ql, rl)
    assert float((qb - rb).abs().max()) <= 2e-6 * max(1.0, float(rb.abs().max()))
    go = torch.randn(B, PC, C, generator=g).to(cuda)
    ql.backward(go)
    rl.backward(go)
    assert float((w.grad - wr.grad).abs().max()) <= 1e-5 * float(wr.grad.abs().max())
    first = w.grad.clone()
    w.grad = None
    ops.cdn_queries(w, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, 0.5, 0.4, NC)[0].backward(go)
    assert torch.equal(first, w.grad)


def test_cls_head_kernels(cuda):
    """ops.global_avg_pool on a channels-last map view and ops.soft_ce_label_smooth against the PyTorch lines they replace
    (mmcls GlobalAveragePooling; LabelSmoothLoss 'original' + soft cross-entropy: slvl_cls_head.py:14-23), values and
    gradients, one-hot and mixed labels."""
    import torch.nn.functional as F
    from rscotr_amd import ops
    g = torch.Generator(device='cpu').manual_seed(2)
    B, H, W, C, K = 2, 16, 16, 768, 45
    tok = torch.randn(B, H * W, C, generator=g).to(cuda).requires_grad_(True)
    ref_tok = tok.detach().clone().requires_grad_(True)
    pooled = ops.global_avg_pool(ops.tokens_to_map(tok, (H, W)))
    ref = ops.tokens_to_map(ref_tok, (H, W)).mean(dim=(2, 3))
    assert pooled.shape == (B, C) and float((pooled - ref).abs().max()) <= 1e-6
    go = torch.randn(B, C, generator=g).to(cuda)
    pooled.backward(go)
    ref.backward(go)
    assert float((tok.grad - ref_tok.grad).abs().max()) <= 1e-7 * float(ref_tok.grad.abs().max()) + 1e-12
    for mixed in (False, True):
        score = (torch.randn(B, K, generator=g) * 3).to(cuda).requires_grad_(True)
        rs = score.detach().clone().requires_grad_(True)
        lab = F.one_hot(torch.randint(0, K, (B,), generator=g), K).float()
        if mixed:
            lab = 0.7 * lab + 0.3 * F.one_hot(torch.randint(0, K, (B,), generator=g), K).float()
        lab = lab.to(cuda)
        loss = ops.soft_ce_label_smooth(score, lab, 0.1, float(B))
        t = lab * 0.9 + 0.1 / K
        rl = (-t * F.log_softmax(rs, dim=-1)).sum() / B
        assert abs(float(loss) - float(rl)) <= 1e-6 * max(1.0, abs(float(rl)))
        (loss * 1.7).backward()
        (rl * 1.7).backward()
        assert float((score.grad - rs.grad).abs().max()) <= 2e-6 * float(rs.grad.abs().max())


@pytest.mark.parametrize('n,used', [(3, (0, 1, 2)), (18, tuple(range(18))), (9, (0, 3, 8)), (4, (2,))])
def test_fan_out_sums_the_consumers_gradients(cuda, n, used):
    """ops.fan_out: n handles of a tensor; the gradient that reaches the tensor is the sum of the consumers' gradients
    (handles without a consumer contribute nothing), summed left to right — identical to autograd's pairwise adds in the
    same order up to rounding, and bit-identical between two runs."""
    from rscotr_amd import ops
    g = torch.Generator(device='cpu').manual_seed(4)
    x = torch.randn(2, 100, 256, generator=g).to(cuda).requires_grad_(True)
    ws = [torch.randn(2, 100, 256, generator=g).to(cuda) for _ in range(n)]
    hs = ops.fan_out(x, n)
    assert len(hs) == n
    sum((hs[i] * ws[i]).sum() for i in used).backward()
    ref = sum(ws[i].double() for i in used)
    assert float((x.grad.double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    first = x.grad.clone()
    x.grad = None
    hs = ops.fan_out(x, n)
    sum((hs[i] * ws[i]).sum() for i in used).backward()
    assert torch.equal(first, x.grad)
