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
