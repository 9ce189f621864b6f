"""Which autograd nodes of one eager iteration are NOT the package's own Functions (they launch device-library kernels in
backward), and where gradients meet (every extra incoming edge of a node output is one element-wise add by the engine):
python scripts/bwd_census.py <task>.  Forward runs under anomaly mode so that every node carries its forward stack."""
import collections, copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rscotr_amd import Config, MODELS, synth, ops
task = sys.argv[1]
CFG = os.path.join(ROOT, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(copy.deepcopy(cfg.model)); model.init_weights(); model.to(dev).train()
batch = synth.make_batch(task, 2, 512, seed=100, device=dev)
b = dict(batch, img_metas=[dict(m) for m in batch['img_metas']])
FREE = ('ViewBackward', 'UnsafeViewBackward', 'ReshapeAliasBackward', 'TBackward', 'TransposeBackward', 'PermuteBackward',
        'UnsqueezeBackward', 'SqueezeBackward', 'AccumulateGrad', 'AliasBackward', 'DetachBackward', 'AsStridedBackward',
        'ExpandBackward', 'UnbindBackward', 'SplitBackward', 'SplitWithSizesBackward')
with torch.autograd.detect_anomaly(check_nan=False):
    loss = model.train_step(b)['loss']


def where(node):
    tb = node.metadata.get('traceback_') or []
    best = 'other'
    for ln in tb:
        if 'rscotr_amd' in ln and 'File' in ln:
            parts = ln.strip().split('\n')
            f = parts[0].split('rscotr_amd/')[-1].replace('", line ', ':').split(',')[0]
            src = parts[1].strip()[:80] if len(parts) > 1 else ''
            best = f'{f} {src}'
    return best


seen, stack = set(), [loss.grad_fn]
incoming = collections.Counter()
nodes = []
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n); nodes.append(n)
    for nxt, idx in n.next_functions:
        if nxt is not None:
            incoming[(nxt, idx)] += 1
            stack.append(nxt)
agg = collections.Counter()
for n in nodes:
    name = type(n).__name__
    if hasattr(n, '_forward_cls') or name.startswith(FREE):
        continue
    agg[(name, where(n))] += 1
print(f'{task}: {len(nodes)} autograd nodes; built-in nodes that launch kernels:')
for (n, w), c in agg.most_common(70):
    print(f'{c:4d} {n:28s} {w}')
fan = collections.Counter()
for (n, idx), c in incoming.items():
    if c > 1:
        fan[(type(n).__name__, where(n) if not type(n).__name__.startswith('AccumulateGrad') else 'param ' + str(tuple(n.variable.shape)))] += c - 1
print(f'gradient fan-in adds: {sum(fan.values())}')
for (n, w), c in fan.most_common(50):
    print(f'{c:4d} {n:28s} {w}')
