import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops
from rscotr_amd._lib import lib
from rscotr_amd.optim import FlatAdamW
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10880
C, H = 256, 2048
g = torch.Generator().manual_seed(3)
x = torch.randn(M, C, generator=g).to(dev)
ps = [torch.nn.Parameter((torch.randn(s, generator=g) * sc).to(dev)) for s, sc in (((H, C), 0.06), ((H,), 0.5), ((C, H), 0.03), ((C,), 0.5))]
opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
W1, b1, W2, b2 = [p.data for p in ps]
ops.RANGES.begin(dev)
bits = torch.empty(int(lib.rscotr_ffn_h3_bits_words(M, C, H)), dtype=torch.int32, device=dev)
for rep in range(3):
    hid, y = ops.FFN_FUSED.run(x, W1, b1, W2, b2, ops.ACT_RELU, bits, 0, None, True)
    h64 = torch.relu(x.double() @ W1.double().T + b1.double())
    bad = ((hid.double() - h64).abs() > 1e-4) | ~torch.isfinite(hid)
    idx = bad.nonzero()
    print('rep', rep, 'bad hid elements', int(bad.sum()), 'of', bad.numel())
    if len(idx):
        r, c = idx[:, 0], idx[:, 1]
        print(' rows: min', int(r.min()), 'max', int(r.max()), ' row%48 hist', torch.bincount(r % 48, minlength=48).tolist())
        print(' col%128 hist (by 4)', torch.bincount((c % 128) // 4, minlength=32).tolist())
        print(' chunk hist', torch.bincount(c // 128, minlength=16).tolist())
        print(' tiles', torch.unique(r // 48)[:20].tolist())
        print(' sample', [(int(a), int(b), float(hid[a, b]), float(h64[a, b])) for a, b in idx[:6].tolist()])
    y64 = h64 @ W2.double().T + b2.double()
    by = ((y.double() - y64).abs() > 1e-3) | ~torch.isfinite(y)
    print('  bad y', int(by.sum()), 'of', by.numel())
