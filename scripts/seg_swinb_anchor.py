"""Swin-B 1024^2 seg step: product vs fp32 oracle vs fp64 oracle (same injected decisions), per-tensor distances of the worst
tensors: which of the two fp32 evaluations sits away from the fp64 one?  python scripts/seg_swinb_anchor.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from util import build_model, load_model_cfg
from parity import run_step_pair, anchor_report
cuda = torch.device('cuda:0')
cfg, mcfg = load_model_cfg(tiny=False)
mcfg['backbone'].update(embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
mcfg['neck']['in_channels'] = [256, 512, 1024]
mcfg['cls_head']['in_channels'] = 1024
model = build_model(mcfg, seed=7).to(cuda)
out, oout, rec, orec, P = run_step_pair(model, mcfg, 'seg', 1024, seed=31, device=cuda, batch_size=1, fp64=True)
rep = anchor_report(model, P, orec['P64'], orec.get('P64b'))
rep.sort(key=lambda r: -max(r['ep'], r['eo']))
print('loss product', float(out['loss']), 'oracle fp32', float(oout['loss']), 'fp64', float(orec['out64']['loss']) if isinstance(orec.get('out64'), dict) else orec.get('out64'))
print('tensors', len(rep), 'product far (ep > 1e-3):', sum(r['ep'] > 1e-3 for r in rep), 'oracle far (eo > 1e-3):', sum(r['eo'] > 1e-3 for r in rep),
      'band moves (amb > 1e-3):', sum(r['amb'] > 1e-3 for r in rep))
for r in rep[:25]:
    print(f"ep {r['ep']:.2e}  eo {r['eo']:.2e}  amb {r['amb']:.2e}  {r['name']}")
