"""Distribution of the relative gradient differences between the shape-static and the dynamic det path at det800
(tests/test_sizes_gpu.py::test_static_det_equals_dynamic_det_at_size): python scripts/static_dynamic_stats.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from util import build_model, load_model_cfg
from rscotr_amd import synth
cuda = torch.device('cuda:0')
cfg, mcfg = load_model_cfg(tiny=False)
size, bs, mg = 800, 4, 50
model = build_model(mcfg, seed=2).to(cuda)
batch = synth.make_batch('det', bs, size, seed=41, device=cuda, max_gt=mg)
rnd = synth.make_rnd(model, synth.make_batch('det', bs, size, seed=41, max_gt=mg), seed=41, device=cuda)
def run(mode):
    model.bbox_head.static_path = mode
    model.zero_grad(set_to_none=True)
    rec = {}
    out = model.train_step(dict(batch, rnd=rnd, record=rec))
    out['loss'].backward()
    torch.cuda.synchronize()
    return out, rec, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
(o1, r1, g1), (o2, r2, g2), (o3, r3, g3) = run(True), run(False), run(True)
ds = []
for n, g in g1.items():
    if float(g2[n].abs().max()) < 1e-7:
        continue
    ds.append((float((g - g2[n]).norm() / (g2[n].norm() + 1e-12)), float((g - g3[n]).norm() / (g3[n].norm() + 1e-12)), n))
ds.sort(reverse=True)
print('tensors', len(ds), 'outside 1e-3:', sum(d[0] > 1e-3 for d in ds), 'static-vs-static max', max(d[1] for d in ds))
for d, d_ss, n in ds[:30]:
    print(f'{d:.2e} (static twice {d_ss:.1e})  {n}')
mx = max(abs(v - o2['log_vars'][k]) / max(abs(v), 1e-3) for k, v in o1['log_vars'].items())
print('max rel log-var difference', mx)
