"""Where the det step at 256^2 (seed 4) leaves the 1e-3 tier under a GEMM variant (RSCOTR_LIB=...): per-tensor relative L2
distance of the product's gradients from the fp32 oracle's, worst first, the product's and the oracle's matchings, and the
forward outputs per decoder layer (scripts/lab; run on the GPU box from the repo root)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import torch
from parity import run_step_pair, grad_report
from util import build_model, load_model_cfg
cuda = torch.device('cuda:0')
cfg, mcfg = load_model_cfg()
model = build_model(mcfg).to(cuda)
out, oout, rec, orec, P = run_step_pair(model, mcfg, 'det', 256, seed=4, device=cuda)
bad = 0
for (s, i), (r, c) in rec['match'].items():
    o = orec['match']['interm' if s == 0 else f'dec{s - 1}'][i]
    bad += not (torch.equal(torch.from_numpy(r), o['pos_inds']) and torch.equal(torch.from_numpy(c), o['pos_assigned_gt_inds']))
print('matchings differing:', bad, 'of', len(rec['match']))
rows = sorted(grad_report(model, P), key=lambda r: -r[3])
for n, mx, frac, l2 in rows[:40]:
    print(f'{l2:9.2e} l2  {mx:8.2f} x tol  {frac:8.5f} of elements over  {n}')
print('record keys', sorted(rec.keys()), sorted(orec.keys()))
for k in sorted(set(rec) & set(orec)):
    a, b = rec[k], orec[k]
    if torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype.is_floating_point:
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        d = (a - b).abs()
        print(k, tuple(a.shape), 'max abs diff', float(d.max()), 'of max', float(b.abs().max()))
        if a.dim() >= 3 and a.shape[0] <= 8:
            print('   per leading index:', [f'{float(d[i].max()):.2e}' for i in range(a.shape[0])])
print({k: (v, oout['log_vars'][k]) for k, v in out['log_vars'].items() if abs(v - oout['log_vars'][k]) > 1e-5 * max(abs(v), 1e-3)})
