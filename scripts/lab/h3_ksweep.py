"""Fixed and per-k-step cost of the 64 x 64 fp16 split kernel with COLD operands inside a replayed hipGraph (no host in the
loop): 10880 x 256 x K for a sweep of K, operand sets used round-robin.  `python scripts/lab/h3_ksweep.py`"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops  # noqa: E402
from rscotr_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
ops.RANGES.enabled = False
M, N = int(os.environ.get('M', 10880)), int(os.environ.get('N', 256))


def slot_of(x):
    s = ops.RANGES.new_slot(dev)
    lib.call('rscotr_amax_f32', x.data_ptr(), x.shape[0], x.shape[1], x.shape[1], s, torch.cuda.current_stream().cuda_stream)
    return s


def graph_time(fns, reps=6):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for f in fns:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for f in fns:
                f()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


for bk in (0, 1):
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        per = (M * K + 2 * M * N) * 4
        nsets = max(6, min(48, int(900e6 // per)))
        As = [torch.randn(M, K, device=dev) for _ in range(nsets)]
        B = torch.randn((K, N) if bk else (N, K), device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        Rs = [torch.randn(M, N, device=dev) for _ in range(nsets)]
        Cs = [torch.empty(M, N, device=dev) for _ in range(nsets)]
        sA = [slot_of(a) for a in As]
        sB = slot_of(B)
        so = ops.RANGES.new_slot(dev)
        r = dict(M=M, N=N, K=K, bk=bk, nsets=nsets)
        for name, kw in (('plain', dict()), ('bias', dict(bias=bias)), ('bias_resid', dict(bias=bias, resid=True)),
                         ('bias_resid_out', dict(bias=bias, resid=True, amax_out=so))):
            fns = []
            for i in range(nsets):
                k = dict(kw)
                if k.pop('resid', False):
                    k['resid'] = Rs[i]
                fns.append(lambda i=i, k=k: ops.gemm(As[i], B, M, N, K, K, B.shape[1], 0, bk, out=Cs[i], amax_a=sA[i], amax_b=sB, **k))
            r[name] = round(graph_time(fns), 2)
        print(json.dumps(r), flush=True)
        del As, Rs, Cs
