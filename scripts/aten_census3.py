"""Every device-library (non-rscotr) kernel of one eager train iteration of a task, attributed to the ATen op that launched
it and to where that op came from: the innermost rscotr_amd frame of the Python stack (forward, and the Python bodies of the
package's own backward Functions) or, for ops the autograd engine issues itself, the node being evaluated (a direct child
`aten::add` of an evaluate_function event = a gradient fan-in accumulation).  python scripts/aten_census3.py <task>"""
import collections, copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    r.train_iter()
    torch.cuda.synchronize()
agg = collections.Counter(); kt = collections.Counter()
nk = 0
for e in prof.events():
    ks = [k for k in (e.kernels or []) if 'rscotr' not in k.name]
    if not ks:
        continue
    # the innermost op that owns these kernels only (children own theirs)
    p, node = e.cpu_parent, None
    chain = []
    while p is not None:
        chain.append(p.name)
        if p.name.startswith('autograd::engine::evaluate_function: '):
            node = p.name.split(': ', 1)[1]
            break
        p = p.cpu_parent
    fr = [s for s in (e.stack or []) if 'rscotr_amd/' in s]
    if fr:
        where = fr[0].split('rscotr_amd/')[-1][:70]
    elif node is not None:
        direct = len(chain) == 1
        where = f'engine[{node}]' + (' fan-in' if direct and e.name in ('aten::add', 'aten::add_') else ' via ' + (chain[0] if chain else ''))
    else:
        where = 'other: ' + ' < '.join(chain[:2])
    shp = str([s for s in (e.input_shapes or []) if s][:2])[:48]
    key = (where, e.name, shp)
    agg[key] += len(ks); nk += len(ks)
    kt[key] += sum(k.duration for k in ks)
print(f'{task}: {nk} device-library kernels in one eager iteration, {sum(kt.values()) / 1e3:.2f} ms')
for key, c in agg.most_common(120):
    print(f'{c:4d} {kt[key]:7.0f}us  {key[0]:72s} {key[1]:22s} {key[2]}')
