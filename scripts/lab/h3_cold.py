"""Split-product kernels with COLD operands (as inside the step: every launch reads tensors that other kernels wrote long
ago): NSETS operand sets of one shape used round-robin, so that a set has left L2 / Infinity Cache when its turn comes again.
`python scripts/lab/h3_cold.py`"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops  # noqa: E402
from rscotr_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
ops.RANGES.enabled = False


def slot_of(x):
    s = ops.RANGES.new_slot(dev)
    lib.call('rscotr_amax_f32', x.data_ptr(), x.shape[0], x.shape[1], x.shape[1], s, torch.cuda.current_stream().cuda_stream)
    return s


def run(fns, reps=3):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


shapes = [(10880, 256, 256, 0, 0, 'enc proj fwd'), (10880, 256, 256, 0, 1, 'enc proj dx'), (10880, 384, 256, 0, 0, 'enc off|aw'),
          (10880, 2048, 256, 0, 0, 'ffn1'), (10880, 256, 2048, 0, 0, 'ffn2'), (10880, 2048, 256, 0, 1, 'ffn2 dx'),
          (8192, 768, 192, 0, 0, 'swin2 fc1'), (8192, 192, 768, 0, 0, 'swin2 fc2'), (2048, 1536, 384, 0, 0, 'swin3 fc1'),
          (2048, 384, 1536, 0, 0, 'swin3 fc2'), (256, 2048, 10880, 1, 1, 'ffn2 dW')]
for M, N, K, ak, bk, tag in shapes:
    per = (M * K + 2 * M * N) * 4
    nsets = max(4, min(48, int(700e6 // per)))
    As = [torch.randn((K, M) if ak else (M, K), device=dev) for _ in range(nsets)]
    B = torch.randn((K, N) if bk else (N, K), device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    Rs = [torch.randn(M, N, device=dev) for _ in range(nsets)]
    Cs = [torch.empty(M, N, device=dev) for _ in range(nsets)]
    sA = [slot_of(a) for a in As]
    sB = slot_of(B)
    so = ops.RANGES.new_slot(dev)
    lda, ldb = As[0].shape[1], B.shape[1]
    ep = {} if (ak and bk) else dict(bias=bias)
    r = dict(tag=tag, M=M, N=N, K=K, ak=ak, bk=bk, nsets=nsets)
    for name, kw in (('x6', {}), ('h3', 'ab'), ('h3_out', 'abo')):
        for epn, epk in (('plain', {}), ('resid', 'r')):
            if ak and bk and epn == 'resid':
                continue
            fns = []
            for i in range(nsets):
                k = dict(ep)
                if epk:
                    k['resid'] = Rs[i]
                if kw:
                    k.update(amax_a=sA[i], amax_b=sB)
                if kw == 'abo':
                    k.update(amax_out=so)
                fns.append(lambda i=i, k=k: ops.gemm(As[i], B, M, N, K, lda, ldb, ak, bk, out=Cs[i], **k))
            r[f'{name}_{epn}'] = round(run(fns), 1)
    # warm reference: one set
    r['x6_warm'] = round(run([lambda: ops.gemm(As[0], B, M, N, K, lda, ldb, ak, bk, out=Cs[0], **ep)] * 8), 1)
    r['h3_warm'] = round(run([lambda: ops.gemm(As[0], B, M, N, K, lda, ldb, ak, bk, out=Cs[0], amax_a=sA[0], amax_b=sB, **ep)] * 8), 1)
    print(json.dumps(r), flush=True)
    del As, Rs, Cs
