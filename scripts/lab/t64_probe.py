import os, sys
sys.path.insert(0, '/root/repo')
import torch
from rscotr_amd import ops
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
def rel(a, ref): return float((a.cpu().double() - ref).abs().max() / ref.abs().max())
for (M, N, K, ak, bk) in [(2500, 768, 3072, 0, 0), (2500, 768, 3072, 0, 1), (2500, 768, 768, 0, 0), (2500, 768, 768, 0, 1), (2500, 768, 2304, 0, 1),
                          (2048, 768, 3072, 0, 0), (2496, 768, 3072, 0, 0), (2500, 704, 3072, 0, 0), (2500, 768, 3104, 0, 0)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ak else (M, K), generator=g)
    B = torch.randn((K, N) if bk else (N, K), generator=g) * 0.05
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = ((A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t()) + bias.double()) + resid.double()
    errs = {}
    for mode in (0, 3):
        lib.call('rscotr_gemm_set_precision', mode)
        out = ops.gemm(A.to(dev), B.to(dev), M, N, K, A.shape[1], B.shape[1], ak, bk, bias=bias.to(dev), resid=resid.to(dev))
        errs[mode] = rel(out, ref)
        # worst row
        e = (out.cpu().double() - ref).abs().max(1)[0]
        errs[f'row{mode}'] = int(e.argmax())
    lib.call('rscotr_gemm_set_precision', 3)
    print(M, N, K, ak, bk, errs)
