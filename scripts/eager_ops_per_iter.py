"""ATen ops the HOST issues per replayed iteration (everything outside the hipGraph): python scripts/eager_ops_per_iter.py"""
import collections, copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from rscotr_amd import Config, MODELS
from rscotr_amd.data import build_synthetic_multidataloader
from rscotr_amd.runner import build_runner
CFG = os.path.join(ROOT, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(copy.deepcopy(cfg.model)); model.init_weights(); model.to(dev).train()
loader = build_synthetic_multidataloader(cfg, dev, size=512, batch_size=2, rank=0)
runner = build_runner(model, cfg, loader)
with runner.on_stream():
    for _ in range(9):
        runner.train_iter()
    torch.cuda.synchronize()
    for _ in range(3):
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            runner.train_iter()
            torch.cuda.synchronize()
        agg = collections.Counter()
        n = 0
        for e in prof.events():
            ks = e.kernels or []
            if ks and e.name.startswith('aten::') or e.name in ('Memcpy', ):
                agg[(e.name, str([s for s in (e.input_shapes or []) if s][:2])[:50], ks[0].name[:40] if ks else '')] += len(ks)
                n += len(ks)
        print(f'== {runner.last_task}: {n} device activities launched by host-side ATen ops')
        for k, c in agg.most_common(30):
            print(f'   {c:3d}  {k[0]:22s} {k[1]:52s} {k[2]}')
