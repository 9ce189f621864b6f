import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
shapes = [(512, 768), (512, 2304), (2048, 384), (2048, 1152), (8192, 576), (32768, 96), (32768, 288), (32768, 48), (200, 256), (37, 45)]
ts = [torch.randn(s, device=dev) * (i + 1) for i, s in enumerate(shapes)]
rows, first = [], 0
slots = []
for t in ts:
    s = ops.RANGES.new_slot(dev)
    slots.append(s)
    rows.append((t.data_ptr(), t.shape[0], t.shape[1], t.shape[1], s, first))
    first += max(1, min(128, t.numel() // 65536))
tab = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
print('table', tab.shape, first, flush=True)
lib.call('rscotr_amax_group', tab.data_ptr(), len(rows), first, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for t, s in zip(ts, slots):
    got = ops.RANGES.word(s)[0]
    print(tuple(t.shape), got, float(t.abs().max()), got == float(t.abs().max()))
