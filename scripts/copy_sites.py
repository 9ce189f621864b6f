"""Python call sites of Tensor.copy_ / clone / contiguous / .to / torch.empty_like-free copies during one eager iteration
(forward AND the Python side of backward): python scripts/copy_sites.py <task>."""
import collections, copy, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rscotr_amd import Config, MODELS, synth
from rscotr_amd.optim import build_optimizer
from rscotr_amd.runner import IterBasedRunner
task = sys.argv[1]
CFG = os.path.join(ROOT, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(copy.deepcopy(cfg.model)); model.init_weights(); model.to(dev).train()
opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config)
batches = [synth.make_batch(task, 2, 512, seed=100 + i, device=dev) for i in range(2)]
class Loop:
    def __iter__(self):
        i = 0
        while True:
            b = batches[i % 2]; i += 1
            yield dict(b, img_metas=[dict(m) for m in b['img_metas']])
r = IterBasedRunner(model, opt, Loop(), graph_tasks=())
for _ in range(3): r.train_iter()
torch.cuda.synchronize()
agg = collections.Counter()
def wrap(name, fn):
    def w(self, *a, **k):
        if isinstance(self, torch.Tensor) and self.is_cuda and not (name == 'contiguous' and self.is_contiguous()):
            where = 'other'
            for fr in reversed(traceback.extract_stack()[:-1]):
                if 'rscotr_amd' in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {(fr.line or '')[:80]}"
                    break
            agg[(name, where, tuple(self.shape))] += 1
        return fn(self, *a, **k)
    return w
for n in ('copy_', 'clone', 'contiguous', 'to', 'float', 'zero_', 'fill_'):
    setattr(torch.Tensor, n, wrap(n, getattr(torch.Tensor, n)))
r.train_iter()
torch.cuda.synchronize()
for (n, w, s), c in agg.most_common(50):
    print(f'{c:4d} {n:11s} {str(s):22s} {w}')
