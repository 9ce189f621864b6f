import sys, torch
sys.path.insert(0, '/root/repo')
from rscotr_amd import ops
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
for (B, H, Lq, Lk, mode) in [(1, 8, 100, 100, 0), (2, 8, 100, 100, 0), (1, 8, 100, 1024, 2), (1, 8, 100, 4096, 2), (2, 8, 100, 4096, 2), (1, 8, 100, 16384, 2), (2, 8, 800, 800, 1), (1, 8, 100, 4096, 3)]:
    S = (torch.randn(B, H, Lq, Lk, generator=g) * 3).to(dev)
    if mode == 0:
        mask = None; mfull = None
    elif mode == 1:
        mask = (torch.rand(Lq, Lk, generator=g) < 0.5).to(dev); mfull = mask[None, None]
    elif mode == 2:
        mask = (torch.rand(B, Lq, Lk, generator=g) < 0.6).to(dev); mfull = mask[:, None]
    else:
        mask = (torch.rand(B * H, Lq, Lk, generator=g) < 0.6).to(dev); mfull = mask.view(B, H, Lq, Lk)
    scale = 32 ** -0.5
    ref = S.double() * scale
    if mfull is not None:
        ref = ref.masked_fill(mfull, float('-inf'))
    ref = ref.softmax(-1)
    P = S.clone()
    lib.call('rscotr_softmax_mask_fwd', P.data_ptr(), 0 if mask is None else mask.data_ptr(), mode, B, H, Lq, Lk, float(scale), ops._stream())
    e_f = float((P.double() - ref).abs().max())
    dP = torch.randn(B, H, Lq, Lk, generator=g).to(dev)
    refd = scale * ref * (dP.double() - (ref * dP.double()).sum(-1, keepdim=True))
    d = dP.clone()
    lib.call('rscotr_softmax_bwd', P.data_ptr(), d.data_ptr(), B * H * Lq, Lk, float(scale), ops._stream())
    e_b = float((d.double() - refd).abs().max() / refd.abs().max())
    print((B, H, Lq, Lk, mode), 'fwd err', e_f, 'bwd rel err', e_b, flush=True)
