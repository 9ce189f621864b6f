"""Does the 64 x 64 fp16 split kernel pay for a COLD instruction cache inside the step?  The same 10880 x 256 x 256 product (bias +
residual, cold operands round-robin, replayed hipGraph) back to back (MODE=a) and with three launches of OTHER kernels with large
code between two of them (MODE=b: a LayerNorm, an fp32-pipe product, a six-term bf16 product, each spread over every CU).  Run
each mode under `rocprofv3 --kernel-trace --stats` and compare the average duration of gemm_h3_kernel (scripts/lab/icache_probe.sh)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops  # noqa: E402
from rscotr_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
ops.RANGES.enabled = False
mode = os.environ.get('MODE', 'a')
M, N, K = 10880, 256, int(os.environ.get('K', 256))


def slot_of(x):
    s = ops.RANGES.new_slot(dev)
    lib.call('rscotr_amax_f32', x.data_ptr(), x.shape[0], x.shape[1], x.shape[1], s, torch.cuda.current_stream().cuda_stream)
    return s


nsets = 24
As = [torch.randn(M, K, device=dev) for _ in range(nsets)]
B = torch.randn(N, K, device=dev) * 0.05
bias = torch.randn(N, device=dev)
Rs = [torch.randn(M, N, device=dev) for _ in range(nsets)]
Cs = [torch.empty(M, N, device=dev) for _ in range(nsets)]
sA = [slot_of(a) for a in As]
sB = slot_of(B)
# the other kernels: every CU gets a workgroup or two, the work is small
Mt = 16384
xt = torch.randn(Mt, 256, device=dev)
g, b_ = torch.ones(256, device=dev), torch.zeros(256, device=dev)
a32 = torch.randn(Mt, 64, device=dev)
w32 = torch.randn(64, 64, device=dev)
c32 = torch.empty(Mt, 64, device=dev)
a6 = torch.randn(Mt, 256, device=dev)
w6 = torch.randn(128, 256, device=dev) * 0.05
c6 = torch.empty(Mt, 128, device=dev)

fns = []
for i in range(nsets):
    fns.append(lambda i=i: ops.gemm(As[i], B, M, N, K, K, K, 0, 0, out=Cs[i], amax_a=sA[i], amax_b=sB, bias=bias, resid=Rs[i]))
    if mode == 'b':
        fns.append(lambda: ops.layer_norm(xt, g, b_))
        fns.append(lambda: ops.gemm(a32, w32, Mt, 64, 64, 64, 64, 0, 0, out=c32))       # fp32 pipe
        fns.append(lambda: ops.gemm(a6, w6, Mt, 128, 256, 256, 256, 0, 0, out=c6))     # six-term bf16 product (no range words)

st = torch.cuda.Stream()
with torch.cuda.stream(st), torch.no_grad():
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for f in fns:
            f()
    for _ in range(8):
        gr.replay()
    torch.cuda.synchronize()
print('done', mode, flush=True)
