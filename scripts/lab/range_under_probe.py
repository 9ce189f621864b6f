"""What too-SMALL range words do to a step (profiles/r6_range_words.txt): a few eager det / seg iterations, are the losses and the
gradient norm finite?  Run with RSCOTR_LIB pointing at a -DRSCOTR_RANGE_UNDER=n build and with the tree's library."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rscotr_amd import Config, MODELS, synth
from rscotr_amd.optim import build_optimizer
from rscotr_amd.runner import IterBasedRunner
CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
for task in ('det', 'seg'):
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
    for it in range(3):
        out = r.train_iter()
        torch.cuda.synchronize()
        nonfinite = sum(int((~torch.isfinite(p.grad)).any()) for p in model.parameters() if p.grad is not None)
        pn = sum(int((~torch.isfinite(p.data)).any()) for p in model.parameters())
        print(f'{task} iteration {it}: outputs {({k: (float(v) if torch.is_tensor(v) and v.numel() == 1 else None) for k, v in (out or {}).items()} if isinstance(out, dict) else out)}; '
              f'parameter tensors with a non-finite entry: {pn}', flush=True)
    opt.close() if hasattr(opt, 'close') else None
