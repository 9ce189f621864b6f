"""The encoder FFN pair as one launch (rscotr_ffn_h3) against the two products it replaces, COLD operands (operand sets used
round-robin so that nothing is L2 / Infinity-Cache resident when its turn comes), forward pair and backward pair.
`python scripts/lab/ffn_cold.py [M]`"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops  # noqa: E402
from rscotr_amd._lib import lib  # noqa: E402
from rscotr_amd.optim import FlatAdamW  # noqa: E402

dev = torch.device('cuda:0')
only_fused = 'fused' in sys.argv[1:]
FLUSH = 'flush' in sys.argv[1:]  # a 768 MB fill between any two calls: weight planes, too, come from HBM (as they do in the step)
args = [a for a in sys.argv[1:] if a not in ('fused', 'flush')]
M = int(args[0]) if args else 10880
C = int(args[1]) if len(args) > 1 else 256
H = int(args[2]) if len(args) > 2 else (2048 if C == 256 else 4 * C)
GELU = C != 256
ACT = ops.ACT_GELU if GELU else ops.ACT_RELU


big = torch.empty(192 << 20, dtype=torch.float32, device=dev) if FLUSH else None


def run(fns, reps=3):
    if FLUSH:
        fns = [(lambda f=f: (big.zero_(), f())) for f in fns]
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


ps = [torch.nn.Parameter(torch.randn(s, device=dev) * sc) for s, sc in (((H, C), 0.06), ((H,), 0.5), ((C, H), 0.03), ((C,), 0.5))]
opt = FlatAdamW([dict(name=f'p{i}', param=p, lr=1e-3, weight_decay=0.0) for i, p in enumerate(ps)])
W1, b1, W2, b2 = [p.data for p in ps]
nsets = 6
xs = [torch.randn(M, C, device=dev) for _ in range(nsets)]
gs = [torch.randn(M, C, device=dev) for _ in range(nsets)]
ops.RANGES.begin(dev)
for t in xs + gs:
    ops.RANGES.of(t, M, C, C)
bits_f = [torch.empty((M, H), dtype=torch.float32, device=dev) if GELU else torch.empty(int(lib.rscotr_ffn_h3_bits_words(M, C, H)), dtype=torch.int32, device=dev) for _ in range(nsets)]
bits_u = [torch.empty(max(M * H // 64, 1), dtype=torch.int64, device=dev) for _ in range(nsets)]
hids = [None] * nsets


def fwd_fused(i):
    hids[i], _ = ops.FFN_FUSED.run(xs[i], W1, b1, W2, b2, ACT, bits_f[i], 0, xs[i], False)


def bwd_fused(i):
    ops.FFN_FUSED.run(gs[i], W2, None, W1, None, ACT, bits_f[i], 1, gs[i], False)


def fwd_unfused(i):
    if GELU:
        h = ops.gemm(xs[i], W1, M, H, C, C, C, 0, 0, bias=b1, act=ops.ACT_GELU, pre=bits_f[i])
    else:
        h = ops.gemm(xs[i], W1, M, H, C, C, C, 0, 0, bias=b1, act=ops.ACT_RELU_BITS, pre=bits_u[i])
    ops.gemm(h, W2, M, C, H, H, H, 0, 0, bias=b2, resid=xs[i], range_out=False)


def bwd_unfused(i):
    if GELU:
        dh = ops.gemm(gs[i], W2, M, H, C, C, H, 0, 1, act=ops.ACT_GELU_GRAD, aux=bits_f[i])
    else:
        dh = ops.gemm(gs[i], W2, M, H, C, C, H, 0, 1, act=ops.ACT_RELU_GRAD_BITS, aux=bits_u[i])
    ops.gemm(dh, W1, M, C, H, H, C, 0, 1, resid=gs[i], range_out=False)


res = dict(M=M, C=C, H=H, nsets=nsets)
relu_ok = GELU or ops.RELU_BITS.ok(M, H, C, C)
res['relu_bits_unfused'] = bool(relu_ok)
for name, f in (('fwd_fused', fwd_fused), ('bwd_fused', bwd_fused), ('fwd_unfused', fwd_unfused), ('bwd_unfused', bwd_unfused)):
    if 'unfused' in name and (not relu_ok or only_fused):
        continue
    res[name + '_us'] = round(run([lambda i=i: f(i) for i in range(nsets)]), 1)
    res[name + '_warm_us'] = round(run([lambda: f(0)] * 6), 1)
print(json.dumps(res), flush=True)
