"""Pre-split weight planes against the in-kernel split, per shape, HOT (same operands back to back) and COLD (a 640 MB
fill between launches: operands come from HBM, as they mostly do inside the step).  python scripts/bench_wplanes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rscotr_amd import ops
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
lib.rscotr_gemm_set_precision(3)
SHAPES = [(10880, 2048, 256), (10880, 256, 2048), (10880, 256, 256), (32768, 384, 96), (32768, 96, 384), (8192, 768, 192),
          (8192, 192, 768), (2048, 1536, 384), (2048, 384, 1536), (2048, 384, 384)]
if os.environ.get('LONGK'):
    SHAPES = [(10880, 256, 2048), (8192, 192, 768), (2048, 384, 1536), (2048, 384, 1152), (1600, 256, 2048), (512, 768, 3072),
              (512, 768, 2304), (2048, 384, 768)]
flush = torch.empty(160 * 1024 * 1024, device=dev)
s = torch.cuda.current_stream().cuda_stream


def timeit(fn, cold, n=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        if cold:
            flush.fill_(float(i))
        ev[i][0].record(); fn(); ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[2:])
    return t[len(t) // 2]


for M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
    npad = (N + 255) // 256 * 256
    planes = torch.empty(npad * K * 3, dtype=torch.int16, device=dev)
    table = torch.from_numpy(np.asarray([[W.data_ptr(), planes.data_ptr(), N, K, K, npad, 0, 0]], dtype=np.int64)).to(dev)
    lib.call('rscotr_gemm_split_weights', table.data_ptr(), 1, (npad * (K // 16) + 255) // 256, s)
    nws = lib.rscotr_gemm_f32_wplanes_workspace(M, N, K)
    ws = torch.empty(max(nws, 4) // 4, device=dev)
    def new():
        lib.call('rscotr_gemm_f32_wplanes', A.data_ptr(), planes.data_ptr(), npad, out.data_ptr(), M, N, K, K, N, bias.data_ptr(),
                 0, 0, 0, 0, 0, 0, 0, 0, ws.data_ptr(), nws, s)
    ops.WPLANES.enabled = False
    def old():
        ops.gemm(A, W, M, N, K, K, K, 0, 0, out=out, bias=bias)
    old(); new()
    r = [timeit(old, False), timeit(new, False), timeit(old, True), timeit(new, True)]
    print(f'M={M:6d} N={N:5d} K={K:5d}  hot: in-kernel {r[0]:7.1f} us  planes {r[1]:7.1f} us   cold: in-kernel {r[2]:7.1f} us  planes {r[3]:7.1f} us',
          flush=True)
