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


def test_seg_forward_head_matches_reference_vectors(cuda):
    """The product's Mask2FormerHead.forward_head on the HIP path (LayerNorm, the three-layer mask embedding as MFMA GEMMs,
    the mask logits on the batched GEMM, the fused resize + `sigmoid < 0.5` attention-mask kernel) against what the
    REFERENCE'S OWN forward_head returned for the same seeded weights and inputs
    (models/multi/seg_head/mask2former_head.py:111-137, scheme 2; tests/golden/make_reference_golden.py)."""
    import types
    import torch.nn as nn
    from rscotr_amd.seg_head import Mask2FormerHead
    z = _load('reference_static.npz')
    C = z['fh_norm_w'].shape[0]
    norm = nn.LayerNorm(C)
    embed = nn.Sequential(nn.Linear(C, C), nn.ReLU(), nn.Linear(C, C), nn.ReLU(), nn.Linear(C, C))
    with torch.no_grad():
        norm.weight.copy_(torch.from_numpy(z['fh_norm_w']))
        norm.bias.copy_(torch.from_numpy(z['fh_norm_b']))
        for i in (0, 2, 4):
            embed[i].weight.copy_(torch.from_numpy(z[f'fh_w{i}']))
            embed[i].bias.copy_(torch.from_numpy(z[f'fh_b{i}']))
    heads = int(z['fh_heads'])
    head = types.SimpleNamespace(transformer_decoder=types.SimpleNamespace(post_norm=norm.to(cuda)), mask_embed=embed.to(cuda),
                                 num_heads=heads)
    dec_out = torch.from_numpy(z['fh_dec_out']).transpose(0, 1).contiguous().to(cuda)   # the product is batch-first
    with torch.no_grad():
        mp, am = Mask2FormerHead.forward_head(head, dec_out, torch.from_numpy(z['fh_mask_feature']).to(cuda), (6, 5))
    assert _rel(mp.cpu().numpy(), z['fh_seg_mask']) < 1e-5
    want = z['fh_attn_mask'].reshape(mp.shape[0], heads, mp.shape[1], -1)
    assert (want == want[:, :1]).all()                  # the reference tiles one mask over the heads
    assert np.array_equal(am.cpu().numpy(), want[:, 0])  # bit for bit (no logit within rounding of 0 in these vectors)


def test_mlvl_cls_pooling_schemes_match_reference_vectors(cuda):
    """The product's MlvlClsHead.pre_logits (token projections on the HIP GEMM) against the reference's own
    pre_logits_3 / 5 / 6 / 7 outputs (models/multi/cls_head/mlvl_cls_head.py:88-119)."""
    import types
    import torch.nn as nn
    from rscotr_amd.cls_head import MlvlClsHead
    z = _load('reference_static.npz')
    feats = [torch.from_numpy(z[f'cls_feat{i}']).to(cuda) for i in range(4)]
    for scheme in (3, 5, 6, 7):
        head = types.SimpleNamespace(scheme=scheme)
        if scheme != 3:
            w = torch.from_numpy(z[f'cls_w{scheme}'])
            head.out_proj = nn.Linear(w.shape[1], 1).to(cuda)
            with torch.no_grad():
                head.out_proj.weight.copy_(w)
                head.out_proj.bias.copy_(torch.from_numpy(z[f'cls_b{scheme}']))
        head._token_proj = types.MethodType(MlvlClsHead._token_proj, head)
        with torch.no_grad():
            got = MlvlClsHead.pre_logits(head, feats).cpu().numpy()
        assert _rel(got, z[f'cls_token{scheme}']) < 1e-5, scheme
