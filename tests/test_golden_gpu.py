"""The HIP kernels (through the C ABI) against the committed golden vectors of tests/golden/ — outputs of
SciPy / torch primitives on seeded inputs (tests/golden/make_golden.py).  Bit-exact for assignment indices,
1e-3 relative (north star) for floating point; the observed error is noted per check."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30))


def test_device_lsap_matches_scipy_golden(cuda):
    from rscotr_amd import ops
    z = _load('lsap_scipy.npz')
    for pre, ks in (('', [0, 1, 2, 3, 6, 7]), ('t', [0, 3])):   # problems with G <= Q (the matcher's orientation)
        for k in ks:
            c = z[f'{pre}cost{k}']
            Q, g = c.shape
            ld = (g + 31) // 32 * 32
            pad = np.full((1, Q, ld), 1e3, dtype=np.float32)
            pad[0, :, :g] = c
            out = ops.lsap_device(torch.from_numpy(pad).to(cuda), torch.tensor([g], dtype=torch.int32, device=cuda)).cpu().numpy()[0]
            want = np.full(g, -1)
            want[z[f'{pre}col{k}']] = z[f'{pre}row{k}']
            assert np.array_equal(out[:g], want) and (out[g:] == -1).all(), (pre, k)


def test_msda_kernels_match_grid_sample_golden(cuda):
    from rscotr_amd import ops
    z = _load('msda_gridsample.npz')
    shapes = torch.from_numpy(z['shapes']).long()
    starts = torch.cat([torch.zeros(1, dtype=torch.long), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    for strategy in ('tiled', 'sorted', 'scatter'):   # 'tiled' is the shipped default
        with ops.STATE.override(msda_bwd=strategy):
            value, loc, attn = (torch.from_numpy(z[k]).to(cuda).requires_grad_(True) for k in ('value', 'loc', 'attn'))
            out = ops.msda(value, shapes.to(cuda), starts.to(cuda), loc, attn)
            out.backward(torch.from_numpy(z['gout']).to(cuda))
        assert _rel(out.detach().cpu().numpy(), z['out']) < 1e-5
        assert _rel(value.grad.cpu().numpy(), z['gvalue']) < 1e-5
        assert _rel(loc.grad.cpu().numpy(), z['gloc']) < 1e-4
        assert _rel(attn.grad.cpu().numpy(), z['gattn']) < 1e-5


def test_fused_adamw_matches_torch_golden(cuda):
    import torch.nn as nn
    from rscotr_amd.optim import FlatAdamW
    z = _load('adamw_torch.npz')
    ps = [nn.Parameter(torch.from_numpy(z[f'p0_{i}'].copy()).to(cuda)) for i in range(4)]
    groups = [dict(param=p, name=f'p{i}', lr=float(z['lr'][i]), weight_decay=float(z['wd'][i])) for i, p in enumerate(ps)]
    opt = FlatAdamW(groups, betas=(0.9, 0.999), eps=1e-8, grad_clip=dict(max_norm=0.1, norm_type=2))
    for step in range(3):
        opt.zero_grad()
        for i, p in enumerate(ps):
            p.grad.copy_(torch.from_numpy(z[f'g{step}_{i}']).to(cuda))
        opt.mark_live([f'p{i}' for i in range(4)])
        opt.step()
        assert abs(float(opt.grad_norm()) - float(z[f'norm_{step}'])) <= 1e-5 * float(z[f'norm_{step}'])
        for i, p in enumerate(ps):
            assert _rel(p.detach().cpu().numpy(), z[f'p{step + 1}_{i}']) < 1e-6


def test_upsample_ce_matches_torch_golden(cuda):
    from rscotr_amd import ops
    z = _load('upsample_ce_torch.npz')
    logit = torch.from_numpy(z['logit']).to(cuda).requires_grad_(True)
    loss, acc = ops.upsample_ce(logit, torch.from_numpy(z['label']).to(cuda), 255)
    loss.backward()
    assert abs(float(loss) - float(z['loss'])) <= 1e-5 * abs(float(z['loss']))
    assert abs(float(acc) - float(z['acc'])) <= 1e-3
    assert _rel(logit.grad.cpu().numpy(), z['dlogit']) < 1e-4


def test_sine_embed_kernel_matches_reference_vectors(cuda):
    """rscotr_sine_embed4 against the output of the REFERENCE'S OWN gen_sineembed_for_position
    (models/multi/bbox_head/transformer.py:43-76, run in the build container: tests/golden/make_reference_golden.py)."""
    from rscotr_amd import ops
    z = _load('reference_static.npz')
    out = ops.sine_embed4(torch.from_numpy(z['sine4_in']).to(cuda)).cpu().numpy()
    assert np.abs(out - z['sine4_out']).max() < 2e-4  # fp32 sin / cos of arguments up to 2 pi
