"""Hot-loop times of the planes x planes product (rscotr_gemm_pp) against the routes it replaces, through the C ABI, on the
step's shapes: python scripts/bench_pp.py [--iters 30].  Columns: pp with a plain store epilogue, pp with bias + ReLU, the
split pass of the A operand, the shipped route of ops.gemm (RSCOTR_PP off)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rscotr_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--iters', type=int, default=30)
a = ap.parse_args()
dev = torch.device('cuda:0')


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [(10880, 2048, 256, 0, 0), (10880, 2048, 256, 0, 1), (10880, 256, 256, 0, 0), (10880, 256, 256, 0, 1), (10880, 384, 256, 0, 0),
          (10880, 256, 384, 0, 1), (32768, 384, 96, 0, 0), (32768, 96, 384, 0, 0), (32768, 288, 96, 0, 0), (8192, 768, 192, 0, 0),
          (8192, 192, 768, 0, 0), (2048, 1536, 384, 0, 0), (2048, 384, 1536, 0, 0), (2048, 1152, 384, 0, 0), (2048, 256, 10880, 1, 1),
          (384, 1536, 2048, 1, 1), (10880, 256, 2048, 0, 0)]
print('    M     N     K ab |  pp plain  pp bias+relu  relu-grad(aux)  bias+resid  split A | shipped (PP off, bias+relu)   [us]')
for M, N, K, ac, bc in SHAPES:
    g = torch.Generator().manual_seed(1)
    A = torch.randn((K, M) if ac else (M, K), generator=g).to(dev)
    B = torch.randn((K, N) if bc else (N, K), generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, device=dev)
    pa, _ = ops.split_planes(A, A.shape[0], A.shape[1], A.shape[1])
    pb, _ = ops.split_planes(B, B.shape[0], B.shape[1], B.shape[1])
    t_plain = timeit(lambda: ops.gemm_pp(pa, ac, pb, bc, M, N, K, out=out), a.iters)
    t_ep = timeit(lambda: ops.gemm_pp(pa, ac, pb, bc, M, N, K, out=out, bias=bias, act=1), a.iters)
    aux = torch.randn(M, N, generator=g).to(dev)
    t_ax = timeit(lambda: ops.gemm_pp(pa, ac, pb, bc, M, N, K, out=out, act=3, aux=aux), a.iters)
    t_rs = timeit(lambda: ops.gemm_pp(pa, ac, pb, bc, M, N, K, out=out, bias=bias, resid=aux), a.iters)
    t_split = timeit(lambda: ops.split_planes(A, A.shape[0], A.shape[1], A.shape[1]), a.iters)
    ops.PP.enabled = False
    t_old = timeit(lambda: ops.gemm(A, B, M, N, K, A.shape[1], B.shape[1], ac, bc, out=out, bias=bias, act=1), a.iters)
    ops.PP.enabled = True
    tf = 2.0 * M * N * K / t_plain * 1e-6
    print(f'{M:6d} {N:5d} {K:5d} {ac}{bc} | {t_plain:8.1f} {t_ep:12.1f} {t_ax:8.1f} {t_rs:8.1f} {t_split:8.1f} | {t_old:8.1f}    pp {tf:6.1f} TF-eq = {tf / 416.7:.2f}')
