"""Which source lines of rscotr_amd issue device-library (ATen) ops in the FORWARD + loss part of one eager iteration:
python scripts/aten_sites.py <task>.  A TorchDispatchMode logs every aten op that touches a CUDA tensor together with the
innermost rscotr_amd frame of the Python stack (backward runs on the autograd thread and is not seen)."""
import collections, copy, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
from rscotr_amd import Config, MODELS, synth
from rscotr_amd.optim import build_optimizer
task = sys.argv[1]
CFG = os.path.join(ROOT, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(copy.deepcopy(cfg.model)); model.init_weights(); model.to(dev).train()
opt = build_optimizer(model, cfg.optimizer, cfg.optimizer_config)
batch = synth.make_batch(task, 2, 512, seed=100, device=dev)
SKIP = ('aten::view', 'aten::_unsafe_view', 'aten::reshape', 'aten::expand', 'aten::slice', 'aten::select', 'aten::t', 'aten::transpose',
        'aten::permute', 'aten::unsqueeze', 'aten::squeeze', 'aten::detach', 'aten::alias', 'aten::as_strided', 'aten::empty',
        'aten::empty_like', 'aten::empty_strided', 'aten::unbind', 'aten::split', 'aten::split_with_sizes', 'aten::flatten',
        'aten::unflatten', 'aten::_local_scalar_dense', 'aten::is_same_size', 'aten::lift_fresh', 'aten::new_empty', 'aten::narrow',
        'aten::chunk', 'aten::view_as', 'aten::expand_as', 'aten::size', 'aten::stride', 'aten::sym_size', 'aten::movedim')
agg = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if name not in SKIP:
            cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()))
            if cuda or name in ('aten::zeros', 'aten::full', 'aten::arange', 'aten::ones'):
                where = 'other'
                for fr in reversed(traceback.extract_stack()):
                    if 'rscotr_amd' in fr.filename and 'aten_sites' not in fr.filename:
                        where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70] if fr.line else ''}"
                        break
                agg[(name, where)] += 1
        return func(*args, **(kwargs or {}))
for _ in range(2):
    out = model.train_step(dict(batch, img_metas=[dict(m) for m in batch['img_metas']]), opt)
with Log():
    out = model.train_step(dict(batch, img_metas=[dict(m) for m in batch['img_metas']]), opt)
torch.cuda.synchronize()
tot = sum(agg.values())
print(f'{task}: {tot} aten ops on CUDA tensors in forward + loss')
for (n, w), c in agg.most_common(80):
    print(f'{c:4d} {n:28s} {w}')
