import os, sys
sys.path.insert(0, '/root/repo')
import torch
from rscotr_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
M, N, K = 10880, 2048, 256
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).to(dev); B = torch.randn(N, K, generator=g).to(dev)
bias = torch.randn(N, generator=g).to(dev); zb = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev)
pa, _ = ops.split_planes(A, M, K, K); pb, _ = ops.split_planes(B, N, K, K)
for rep in range(2):
    for name, kw in [('plain', {}), ('relu', dict(act=1)), ('bias', dict(bias=bias)), ('zero bias', dict(bias=zb)), ('bias+relu', dict(bias=bias, act=1)), ('gelu', dict(act=2)), ('plain', {})]:
        print(f'{name:10s} {timeit(lambda: ops.gemm_pp(pa, 0, pb, 0, M, N, K, out=out, **kw)):7.1f} us')
