"""Run N eager iterations of ONE task (for per-task rocprofv3 kernel statistics)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rscotr_amd import Config, MODELS, synth
from rscotr_amd.optim import build_optimizer
from rscotr_amd.runner import IterBasedRunner
task, n = sys.argv[1], int(sys.argv[2])
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(copy.deepcopy(cfg.model)); model.init_weights(); model.to(dev).train()
opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config)
batches = [synth.make_batch(task, 2, 512, seed=100 + i, device=dev) for i in range(4)]
class Loop:
    def __iter__(self):
        i = 0
        while True:
            b = batches[i % 4]; i += 1
            yield dict(b, img_metas=[dict(m) for m in b['img_metas']])
r = IterBasedRunner(model, opt, Loop(), graph_tasks=())
for _ in range(3): r.train_iter()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(n): r.train_iter()
torch.cuda.synchronize()
print(f'{task}: {(time.time() - t0) / n * 1e3:.1f} ms/iter', flush=True)
