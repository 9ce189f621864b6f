"""Weight-gradient products (both operands k-major) of the co-training step, one process per environment setting (the loop /
tile knobs of csrc/gemm.hip are read once): median of cold launches (a 640 MB fill before each: operands from HBM), and the
result checked against fp64.  python scripts/lab/dw_lab.py [M,N,K ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rscotr_amd import ops
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
lib.rscotr_gemm_set_precision(3)
shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or \
    [(256, 2048, 10880), (2048, 256, 10880), (384, 1536, 2048), (1536, 384, 2048), (1152, 384, 2048), (256, 2048, 1600)]
flush = torch.empty(160 * 1024 * 1024, device=dev)
out = []
for M, N, K in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
    C = torch.zeros(M, N, device=dev)
    rs = torch.zeros(M, device=dev)
    fn = lambda: ops.gemm(A, B, M, N, K, M, N, 1, 1, out=C, rowsum=rs)
    fn()
    ref = A.double().t() @ B.double()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    rerr = float((rs.double() - A.double().sum(0)).abs().max() / A.double().sum(0).abs().max())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(14)]
    for i, (a, b) in enumerate(ev):
        flush.fill_(float(i)); a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[2:])[6]
    out.append(f'{M}x{N}x{K}: {t:6.1f} us {2e-6 * M * N * K / t:6.1f} TF err {err:.1e}/{rerr:.1e}')
print(' | '.join(out))
