"""Gradients + recorded attention masks of the Swin-B 1024^2 seg step (tests/test_sizes_gpu.py) under the current
environment -> a .pt file; with two files: compare.  python scripts/seg_swinb_dump.py out.pt | compare a.pt b.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
if sys.argv[1] == 'compare':
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    print('loss', a['loss'], b['loss'])
    for i, (ma, mb) in enumerate(zip(a['masks'], b['masks'])):
        print('mask', i, tuple(ma.shape), 'bits differing', int((ma != mb).sum()), 'of', ma.numel(), 'blocked frac', float(ma.float().mean()))
    rows = []
    for n in a['grads']:
        ga, gb = a['grads'][n].double(), b['grads'][n].double()
        rows.append((float((ga - gb).norm() / (gb.norm() + 1e-30)), n))
    for d, n in rows:
        if 'transformer_decoder' in n or d > 1e-3:
            print(f'{d:.2e} {n}')
    sys.exit(0)
from util import build_model, load_model_cfg
from rscotr_amd import synth
cuda = torch.device('cuda:0')
cfg, mcfg = load_model_cfg(tiny=False)
mcfg['backbone'].update(embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
mcfg['neck']['in_channels'] = [256, 512, 1024]
mcfg['cls_head']['in_channels'] = 1024
model = build_model(mcfg, seed=7).to(cuda)
batch_cpu = synth.make_batch('seg', 1, 1024, seed=31)
batch = synth.make_batch('seg', 1, 1024, seed=31, device=cuda)
rnd = synth.make_rnd(model, batch_cpu, seed=31, device=cuda)
rec = {}
out = model.train_step(dict(batch, rnd=rnd, record=rec))
out['loss'].backward()
torch.cuda.synchronize()
torch.save(dict(loss=float(out['loss']), masks=[m.cpu() for m in rec['attn_masks']],
                grads={n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}), sys.argv[1])
